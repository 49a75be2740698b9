"""MI355X-native any-hit shadow-ray path behind the data contract of kayru/RayTracedShadows.

The product is ``librts.so`` (C ABI in ``include/rts.h``: host BVH producer + hand-written HIP
traversal kernels for gfx950).  This package is only the ctypes view of that ABI -- the same
names the reference uses (``BVHBuilder.build``, ``m_nodes``, ``m_packedNodes``,
``RayTracingConstants``) -- plus the harness that synthesises scenes, cameras and G-buffer
positions.  There is no CPU fallback: touching :mod:`raytracedshadows_amd.api` raises if the
library is missing (``raytracedshadows_amd.build.build_product()`` builds it), and tracing raises if
there is no GPU.
"""
_API = ("BVHBuilder", "BVHNode_dtype", "Light", "RayTracingConstants", "RtsError", "ShadowContext",
        "bvh_validate", "device_count", "lib_path", "packed_count")

__all__ = list(_API)


def __getattr__(name):
    # resolved on first use so that `raytracedshadows_amd.build` can be imported before librts.so exists
    if name in _API:
        from . import api
        return getattr(api, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
