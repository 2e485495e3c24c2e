#!/usr/bin/env python3
"""Within-process interleaved A/B of the GEMM kernel variants on the fine-tune step's shapes (guide rule 24).
Usage (GPU box): python tools/gemm_bench.py > gpurun_out/gemm_variants.md"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

SHAPES = [(1184, 4096, 22016), (1184, 22016, 4096), (1184, 4096, 12288), (1184, 4096, 11008), (1184, 12288, 4096),
          (1184, 11008, 4096), (1184, 4096, 4096), (2056, 1408, 6144), (2056, 6144, 1408), (2056, 4224, 1408),
          (2056, 1408, 1408), (648, 768, 3072), (648, 3072, 768), (8192, 8192, 8192)]
VARIANTS = {1: "2st/128/k64 (2blk)", 4: "2st/64/k64 (3blk)", 6: "2st/128/k32 (4blk)", 7: "3st/128/k32 (3blk)", 8: "4st/128/k32 (2blk)"}
ROUNDS, INNER = 5, 10


def main():
    dev = "cuda"
    print("| M | N | K | " + " | ".join(VARIANTS.values()) + " |")
    print("|---|---|---|" + "---|" * len(VARIANTS))
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = None
        best = {v: 1e9 for v in VARIANTS}
        for v in VARIANTS:   # warm + correctness cross-check
            ops.gemm(a, b, out=out, variant=v)
            if ref is None:
                ref = out.float().clone()
            else:
                err = (out.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
                assert err < 2e-2, (M, N, K, v, err)
        for _ in range(ROUNDS):
            for v in VARIANTS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(INNER):
                    ops.gemm(a, b, out=out, variant=v)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / INNER)
        fl = 2.0 * M * N * K
        print(f"| {M} | {N} | {K} | " + " | ".join(f"{fl / (best[v] * 1e-3) / 1e12:.0f} TF ({best[v] * 1e3:.0f} us)" for v in VARIANTS) + " |")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
