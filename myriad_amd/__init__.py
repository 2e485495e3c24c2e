"""myriad_amd -- MI355X-native (gfx950) hot path of tzjtatata/Myriad behind the reference's model-class API.

Python host code (this package) calls hand-written HIP kernels in libmyriad_hip.so through the C ABI declared in
include/myriad_hip.h.  PyTorch supplies device memory, streams and torch.distributed (RCCL) only.
"""
import os as _os

# Kernel arguments in device memory: HIP's default places the kernarg segment in host memory, and every kernel begins by fetching
# it over the host link -- 0.9 ms of a 42 ms step, 1.5 ms of the 20 ms batch-1 step (profiles/r04_gemm_x4.md).  Must be set before
# the HIP runtime initialises (first device call); a value the user set is respected.
_kernarg_user_set = "HIP_FORCE_DEV_KERNARG" in _os.environ
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def _runtime_already_up() -> bool:
    import sys as _sys
    t = _sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# True when this import came too late for the setting to take (the host framework had already touched the device): the step
# then runs ~2 % slower and _lib.load() says so once.  Set HIP_FORCE_DEV_KERNARG=1 in the job's environment to be sure.
KERNARG_SET_TOO_LATE = (not _kernarg_user_set) and _runtime_already_up()

__version__ = "0.1.0"
