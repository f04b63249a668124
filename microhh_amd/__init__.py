"""microhh_amd -- MI355X-native RHS + pressure hot path of MicroHH behind a C ABI (include/mhh_hip.h).

Compute lives in microhh_amd/libmhh_hip.so (hand-written HIP for gfx950, built by `python -m microhh_amd.build`).
There is no CPU fallback: using the package without that library raises.
"""
from . import capi, grid  # noqa: F401

__all__ = ["capi", "grid"]
