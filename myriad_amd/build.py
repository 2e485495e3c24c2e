"""Build libmyriad_hip.so (gfx950) in-tree with hipcc.  `python -m myriad_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmyriad_hip.so")
SOURCES = ["gemm", "gemm_256", "gemm_x4", "gemv", "attention", "attn_seq", "norm", "elementwise", "conv", "loss", "lowrank", "lora", "expert", "image", "selfsup", "optim", "prof", "ctx", "version"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h")]
    # the four-wave GEMM's K loop is written by a generator (the schedule lives there); regenerate when it is newer
    gen, inc = os.path.join(CSRC, "gen_gemm_x4.py"), os.path.join(CSRC, "gemm_x4_loop.inc")
    if force or _stale(inc, [gen]):
        subprocess.check_call([sys.executable, gen])
    hdrs.append(inc)

    def one(name):
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(OBJ, name + ".o")
        if force or _stale(obj, [src, *hdrs]):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
