R=$(pwd); O=$R/gpurun_out/r4a; mkdir -p $O
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/vk -o vk --output-format csv -- python $R/tools/vendor_kernel_names.py > $O/vk.log 2>&1
cd $R
find $O/vk -name "*kernel_stats.csv" | head -1 | xargs cat > $O/vendor_kernel_stats.csv
find $O/vk -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/vendor_kernel_trace.csv
rm -rf $O/vk
python tools/gemm_vendor_calib.py > $O/vendor_calib.md 2>&1
head -c 1500 $O/bench.json; echo; cat $O/vendor_calib.md; cut -c1-400 $O/vendor_kernel_stats.csv | head
