"""ctypes binding of libmyriad_hip.so.  The product path has NO fallback: if the HIP library is missing or a
symbol declared in include/myriad_hip.h is not exported, importing/using the ops raises loudly."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
# MYRIAD_HIP_DEBUG_LIB=1 (set by the tools/ scripts that need a timing probe or a sweep switch) loads the -DMH_DEBUG_HOOKS build
LIB_PATH = os.path.join(_PKG, "libmyriad_hip_dbg.so" if os.environ.get("MYRIAD_HIP_DEBUG_LIB") == "1" else "libmyriad_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "myriad_hip.h")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
       "unsigned long long": ctypes.c_ulonglong}


class MyriadHipError(RuntimeError):
    pass


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    """Parse the C ABI header into {name: (restype, [argtypes])} so the binding can never drift from it."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|long|double|const char\*)\s+(mh_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        restype = {"int": ctypes.c_int, "long": ctypes.c_long, "double": ctypes.c_double, "const char*": ctypes.c_char_p}[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("mh_stream_t"):
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = a.replace("const ", "").split()
                    base = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]
                    argtypes.append(_CT[base])
        out[name] = (restype, argtypes)
    return out


_lib = None
_sigs = None


def signatures():
    global _sigs
    if _sigs is None:
        _sigs = parse_header()
    return _sigs


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MyriadHipError(
            f"{LIB_PATH} not found. The MI355X HIP library is required (no CPU / eager fallback exists). "
            "Build it with `python -m myriad_amd.build` (hipcc --offload-arch=gfx950).")
    # The host framework's HIP runtime first: PyTorch-ROCm ships its own libamdhip64 and the process must hold ONE runtime --
    # loading this library before torch pulls in /opt/rocm's copy, and kernels registered with one runtime cannot be launched on
    # the other's streams (seen as MH_ERR_LAUNCH at the first launch when build() ran before smoke() in one process).
    import torch  # noqa: F401
    import myriad_amd
    if myriad_amd.KERNARG_SET_TOO_LATE:
        import warnings
        warnings.warn("myriad_amd was imported after the HIP runtime initialised: HIP_FORCE_DEV_KERNARG=1 did not take, kernel "
                      "arguments stay in host memory (~2 % per step, more at batch 1).  Export it in the job's environment.",
                      RuntimeWarning, stacklevel=2)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in signatures().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MyriadHipError(f"libmyriad_hip.so does not export {name} declared in include/myriad_hip.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_ERR = {-1: "MH_ERR_ARG (bad dims/alignment)", -2: "MH_ERR_LAUNCH (hip launch failed)", -3: "MH_ERR_UNSUPPORTED"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MyriadHipError(f"{what} failed: {_ERR.get(rc, rc)}")
