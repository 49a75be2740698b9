"""MI355X-native any-hit shadow-ray path behind the data contract of kayru/RayTracedShadows.

The product is ``librts.so`` (C ABI in ``include/rts.h``: host BVH producer + hand-written HIP
traversal kernels for gfx950).  This package is only the ctypes view of that ABI -- the same
names the reference uses (``BVHBuilder.build``, ``m_nodes``, ``m_packedNodes``,
``RayTracingConstants``) -- plus the harness that synthesises scenes, cameras and G-buffer
positions.  There is no CPU fallback: importing :mod:`raytracedshadows_amd.api` raises if the
library is missing, and tracing raises if there is no GPU.
"""
from .api import (BVHBuilder, BVHNode_dtype, Light, RayTracingConstants, RtsError, ShadowContext,
                  bvh_validate, device_count, lib_path, packed_count)

__all__ = ["BVHBuilder", "BVHNode_dtype", "Light", "RayTracingConstants", "RtsError", "ShadowContext",
           "bvh_validate", "device_count", "lib_path", "packed_count"]
