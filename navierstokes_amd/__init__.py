"""MI355X-native (gfx950) NavierStokes.jl projection-method time step.

`navierstokes_amd.csrc` holds the HIP kernels and the C ABI (libnsk.so, declared in
include/nsk.h); the Python modules are the host-side mirror of the reference's Julia drivers.
"""
from .params import NSParams, ns_params  # noqa: F401

__all__ = ["NSParams", "ns_params"]
