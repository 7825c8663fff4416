"""colormipsearch_amd -- MI355X-native color-depth MIP search core (pixel-match scan + GradientAreaGap).

`capi`  : ctypes binding of libcds_hip.so (include/cds_hip.h), the drop-in C ABI.
`cds`   : host-side mirror of the reference's ColorDepthSearchAlgorithm plugin surface.
`dist`  : target-sharded multi-GPU search (one process per GPU, one RCCL gather).
"""
from . import capi  # noqa: F401

__all__ = ["capi", "cds", "dist"]
