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
# the same sources with -DMH_DEBUG_HOOKS: adds the mhdbg_* timing probes / sweep switches the tools/ scripts use (one of them,
# the read-out without its stores, gives wrong results on purpose); the product and the tests never load it
LIB_DBG = os.path.join(HERE, "libmyriad_hip_dbg.so")
DBG_SOURCES = ["gemm", "gemm_256", "gemm_x4", "attn_seq", "lora"]      # the files that hold a hook
SOURCES = ["gemm", "gemm_256", "gemm_x4", "gemv", "attention", "attn_seq", "attn_full", "norm", "elementwise", "conv", "loss", "lowrank", "lora", "expert", "image", "selfsup", "optim", "prof", "ctx", "version"]
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

    def one(job):
        name, dbg = job
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(OBJ, name + (".dbg.o" if dbg else ".o"))
        if force or _stale(obj, [src, *hdrs]):
            cmd = [hipcc, *FLAGS, *(["-DMH_DEBUG_HOOKS"] if dbg else []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    jobs = [(n, False) for n in SOURCES] + [(n, True) for n in DBG_SOURCES]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        built = dict(zip(jobs, ex.map(one, jobs)))
    objs = [built[(n, False)] for n in SOURCES]
    objs_dbg = [built[(n, n in DBG_SOURCES)] for n in SOURCES]
    for lib, oo in ((LIB, objs), (LIB_DBG, objs_dbg)):
        if force or _stale(lib, oo):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *oo, "-ldl"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
