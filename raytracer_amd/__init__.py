"""raytracer_amd — MI355X (gfx950) implementation of the path-tracing hot path of
VictorOrrios/Raytracer behind a C ABI (include/rt_abi.h).

    capi      ctypes mirror of the C ABI (loads lib/libraytracer_amd.so, no fallback)
    renderer  Scene / Context wrappers
    parallel  multi-GPU partition of a frame (row strips) and of samples (frames) over
              torch.distributed
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
