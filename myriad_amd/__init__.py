"""myriad_amd -- MI355X-native (gfx950) hot path of tzjtatata/Myriad behind the reference's model-class API.

Python host code (this package) calls hand-written HIP kernels in libmyriad_hip.so through the C ABI declared in
include/myriad_hip.h.  PyTorch supplies device memory, streams and torch.distributed (RCCL) only.
"""
__version__ = "0.1.0"
