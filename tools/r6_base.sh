#!/bin/bash
# Round-6 baseline look (through gpurun): r5_quick's set + the ViT alone + the ViT GEMM sweep
R=$(pwd); O=$R/gpurun_out/${1:-r6base}; mkdir -p $O
bash tools/r5_quick.sh ${1:-r6base} > $O/quick.log 2>&1
python tools/vit_alone.py > $O/vit_alone.log 2>&1
python tools/gemm_vit_sweep.py > $O/vit_sweep.log 2>&1
tail -5 $O/quick.log; cat $O/vit_alone.log; cat $O/vit_sweep.log
