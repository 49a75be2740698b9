"""ctypes view of ``librts.so`` (C ABI: ``include/rts.h`` / ``include/rts_scene.h``).

Mirrors the reference's interface for the shadow path:

* :class:`BVHBuilder` -- ``BVHBuilder::build(vertices, stride, indices, primCount)`` with the
  public members ``m_nodes`` / ``m_packedNodes`` (reference ``Source/BVHBuilder.h:27-32``).
* :class:`RayTracingConstants` -- the 64-byte UBO (``Source/RayTracedShadows.h:56-62``).
* :class:`ShadowContext` -- owns the device BVH buffer and issues the dispatch that
  ``RayTracedShadowsApp::renderShadowMaskCompute`` issues (``Source/RayTracedShadows.cpp:570-595``).

No CPU fallback exists here by design: a missing library raises at import, a missing GPU raises
at context creation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # RTS_LIB lets an experiment load an alternative build of the SAME library (e.g. other compiler flags)
    return os.environ.get("RTS_LIB") or os.path.join(_HERE, "librts.so")


class RtsError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.rts_status_string(status).decode() if _lib is not None else "?"
        super().__init__(f"{where}: rts status {status} ({msg})")


def _load():
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the shadow path.")
    return C.CDLL(path)


_lib = None
_lib = _load()

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class RayTracingConstants(C.Structure):
    """``struct RayTracingConstants`` (RayTracedShadows.h:56-62) == UBO ``Constants`` (comp:3-9)."""
    _fields_ = [("cameraPosition", C.c_float * 4), ("cameraDirection", C.c_float * 4),
                ("lightDirection", C.c_float * 4), ("renderTargetSize", C.c_float * 4)]

    @classmethod
    def make(cls, camera_position, light_direction, width, height, camera_direction=(0, 0, -1)):
        k = cls()
        for i in range(3):
            k.cameraPosition[i] = np.float32(camera_position[i])
            k.cameraDirection[i] = np.float32(camera_direction[i])
            k.lightDirection[i] = np.float32(light_direction[i])
        k.renderTargetSize[0] = width
        k.renderTargetSize[1] = height
        k.renderTargetSize[2] = 1.0 / width
        k.renderTargetSize[3] = 1.0 / height
        return k

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class Light(C.Structure):
    """``rts_light``: directional (the reference) / point light, 1..64 samples."""
    _fields_ = [("type", C.c_uint32), ("nsamples", C.c_uint32), ("xyz", C.c_float * 3),
                ("table", C.c_uint32), ("offsets", (C.c_float * 4) * 64)]
    DIRECTIONAL = 0
    POINT = 1

    @classmethod
    def make(cls, kind, xyz, offsets=None, nsamples=None):
        """offsets: the sample offsets (<= 64).  nsamples < len(offsets): PER-PIXEL jitter -- every pixel takes `nsamples`
        consecutive entries of the table from a start hashed from its index (``rts_light.table``, include/rts.h)."""
        lt = cls()
        lt.type = kind
        for i in range(3):
            lt.xyz[i] = np.float32(xyz[i])
        lt.nsamples = 1
        if offsets is not None:
            offsets = np.asarray(offsets, dtype=np.float32)
            lt.nsamples = offsets.shape[0]
            if nsamples is not None and nsamples != offsets.shape[0]:
                lt.nsamples, lt.table = nsamples, offsets.shape[0]
            for j in range(offsets.shape[0]):
                for i in range(3):
                    lt.offsets[j][i] = offsets[j, i]
        return lt


#: numpy view of ``struct BVHNode`` (BVHBuilder.h:8-20)
BVHNode_dtype = np.dtype([("bboxMin", np.float32, 3), ("prim", np.uint32),
                          ("bboxMax", np.float32, 3), ("next", np.uint32)])


def _sig(name, restype, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("rts_status_string", C.c_char_p, C.c_int)
_sig("rts_bvh_packed_count", C.c_size_t, C.c_uint32)
_sig("rts_bvh_node_count", C.c_size_t, C.c_uint32)
_sig("rts_bvh_build", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("rts_bvh_build_ex", C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
     C.c_void_p, C.c_size_t, C.c_void_p)
_sig("rts_bvh_validate", C.c_int, C.c_void_p, C.c_size_t, _u32p)
_sig("rts_bvh_build_device", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint32,
     C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float))
_sig("rts_bvh_build_device_ex", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int,
     C.c_uint32, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float))
_sig("rts_device_count", C.c_int, C.POINTER(C.c_int))
_sig("rts_ctx_create", C.c_int, C.c_int, C.POINTER(C.c_void_p))
_sig("rts_ctx_destroy", C.c_int, C.c_void_p)
_sig("rts_ctx_set_bvh", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_ctx_set_option", C.c_int, C.c_void_p, C.c_char_p, C.c_int)
_sig("rts_ctx_get_option", C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_int))
_sig("rts_trace_shadow_mask", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p,
     C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
_sig("rts_trace_shadow_mask_device", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light),
     C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)
_sig("rts_trace_shadow_mask_stripes_device", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light),
     C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)
_sig("rts_trace_rays", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_sig("rts_trace_rays_device", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
_sig("rts_device_malloc", C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t)
_sig("rts_device_free", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_memcpy_h2d", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_memcpy_d2h", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_stream_synchronize", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_timer_begin", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_timer_end", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_timer_elapsed_ms", C.c_int, C.c_void_p, C.POINTER(C.c_float))
_sig("rts_ctx_last_kernel_name", C.c_char_p, C.c_void_p)
_sig("rts_ctx_set_tile_order", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_ctx_read_wave_stats", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_ctx_read_wave_realtime", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_ctx_read_clock_probe", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_stream_create", C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_sig("rts_stream_destroy", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_ctx_autotune", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
     C.POINTER(C.c_int), C.POINTER(C.c_float))


def split_front_order(life_us, tiles, first_record=0, xcd_square=0, life_block=0):
    """rtsh_split_front_order: the planner's front order (host logic, no device): indices of the tiles in record order."""
    life = np.ascontiguousarray(life_us, np.float32)
    t = np.ascontiguousarray(tiles, np.uint32)
    out = np.zeros(t.size, np.uint32)
    _check(_lib.rtsh_split_front_order(_ptr(life), _ptr(t), t.size, first_record, xcd_square, life_block, _ptr(out)), "rtsh_split_front_order")
    return out


class SplitPlan(C.Structure):
    """rts_split_plan (include/rts.h)."""
    _fields_ = [("min_life_us", C.c_float), ("end_after_us", C.c_float), ("piece_us", C.c_float), ("front_life_us", C.c_float), ("front_share", C.c_float), ("max_pieces", C.c_uint32), ("max_tiles", C.c_uint32),
                ("xcd_square", C.c_uint32), ("life_block", C.c_uint32), ("reserved_", C.c_uint32), ("prev_stats", C.c_void_p), ("prev_realtime", C.c_void_p), ("prev_waves", C.c_size_t)]


_sig("rts_ctx_plan_splits", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p, C.c_uint32, C.c_uint32,
     C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(SplitPlan), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_sig("rts_ctx_plan_splits_stripes", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p, C.c_uint32,
     C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(SplitPlan), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_sig("rts_ctx_clear_splits", C.c_int, C.c_void_p)
_sig("rts_ctx_plan_tile_order", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p, C.c_uint32, C.c_uint32,
     C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32))
_sig("rtsh_split_front_order", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
_sig("rts_selftest_reciprocal", C.c_int, C.c_void_p, C.c_void_p)
_sig("rts_ctx_get_split_plan", C.c_int, C.c_void_p, C.POINTER(SplitPlan))
_sig("rts_ctx_autotune_stripes", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
     C.c_uint32, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float))
_sig("rts_ctx_read_piece_stats", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_sig("rts_timer_mark", C.c_int, C.c_void_p, C.c_void_p, C.c_uint32)
_sig("rts_timer_between_ms", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float))
_sig("rts_device_mem_info", C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
_sig("rtsh_primary_positions", C.c_int, C.c_void_p, C.c_size_t, _f32p, _f32p, C.c_float, C.c_uint32, C.c_uint32,
     C.c_void_p, C.POINTER(C.c_uint64), C.c_int)
_sig("rtsh_primary_gbuffer", C.c_int, C.c_void_p, C.c_size_t, _f32p, _f32p, C.c_float, C.c_uint32, C.c_uint32,
     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int)
_sig("rtsh_primary_gbuffer_device", C.c_int, C.c_void_p, _f32p, _f32p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p,
     C.c_void_p, C.c_void_p)
_sig("rtsh_combine_device", C.c_int, C.c_void_p, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)
_sig("rtsh_combine", C.c_int, C.POINTER(RayTracingConstants), C.POINTER(Light), C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_uint32, C.c_uint32, C.c_void_p)
_sig("rtsh_obj_load", C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, _u32p, _f32p, _f32p)
_sig("rtsh_obj_parse_float", C.c_float, C.c_char_p, C.POINTER(C.c_int))


def _check(status, where):
    if status != 0:
        raise RtsError(status, where)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def packed_count(prim_count):
    """``m_packedNodes.size()`` for ``prim_count`` triangles (= 5P-2)."""
    return int(_lib.rts_bvh_packed_count(prim_count))


def bvh_validate(packed):
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1, 4)
    p = C.c_uint32(0)
    _check(_lib.rts_bvh_validate(_ptr(packed), packed.shape[0], C.byref(p)), "rts_bvh_validate")
    return int(p.value)


def device_count():
    n = C.c_int(0)
    _lib.rts_device_count(C.byref(n))
    return int(n.value)


class BVHBuilder:
    """Same surface as the reference's ``struct BVHBuilder`` (Source/BVHBuilder.h:27-32).

    ``m_nodes``: structured array of ``BVHNode`` (2P-1), ``m_packedNodes``: ``uint32[5P-2, 4]``
    (one row per ``BVHPackedNode`` / GLSL ``vec4``).
    """

    def __init__(self, sah_prim_limit=1000000, threads=0):
        self.m_nodes = np.zeros(0, dtype=BVHNode_dtype)
        self.m_packedNodes = np.zeros((0, 4), dtype=np.uint32)
        self.sah_prim_limit = sah_prim_limit
        self.threads = threads

    def build(self, vertices, stride, indices, primCount):
        """``stride`` is in floats, exactly like the reference (it passes 8)."""
        vertices = np.ascontiguousarray(vertices, dtype=np.float32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        if primCount > 0:
            if indices.size < 3 * primCount:
                raise RtsError(1, "BVHBuilder.build: indices shorter than 3*primCount")
            if vertices.size < (int(indices[:3 * primCount].max()) * stride + 3):
                raise RtsError(1, "BVHBuilder.build: index out of range of the vertex array")
        n = packed_count(primCount)
        packed = np.zeros((max(n, 1), 4), dtype=np.uint32)
        nodes = np.zeros(max(2 * primCount - 1, 1), dtype=BVHNode_dtype)
        st = _lib.rts_bvh_build_ex(_ptr(vertices), stride, _ptr(indices), primCount, self.sah_prim_limit,
                                   self.threads, _ptr(packed), packed.shape[0], _ptr(nodes))
        _check(st, "rts_bvh_build")
        self.m_packedNodes = packed[:n]
        self.m_nodes = nodes[:2 * primCount - 1]
        return self


def bvh_build_device(ctx, vertices, stride, indices, prim_count, install=False, want_packed=True, algorithm="sah",
                     radius=0):
    """BVH build on the GPU.  "sah" (default): BVHBuilder's own split rule for every node, level by level on the device
    -- ``radius`` is then the range size above which the median split is used, 0 = the reference's 1 000 000.  Over the
    Morton order: "ploc" (locally-ordered clustering, ``radius`` neighbours each way, 0 = 16), "lbvh" (Karras hierarchy),
    "ploc_sah" (PLOC below 65 536 clusters, full-sweep SAH over the clusters on the host above).
    ``vertices`` may be ``(device pointer, number of floats)`` and ``indices`` a device pointer: geometry that already lives on
    the context's device is used in place.  Returns (packed or None, device milliseconds)."""
    import numbers

    def device_pointer(v):                           # a plain or numpy integer, or a ctypes pointer value
        if isinstance(v, C.c_void_p):
            v = v.value
        if isinstance(v, numbers.Integral) and not isinstance(v, bool) and int(v) > 0:
            return int(v)
        return None

    if isinstance(vertices, tuple):                  # (device pointer, number of floats): geometry already on the device
        if len(vertices) != 2 or device_pointer(vertices[0]) is None or int(vertices[1]) < 3 * stride:
            raise ValueError("vertices: expected (device pointer, number of floats >= 3 * stride)")
        v_ptr, v_floats = C.c_void_p(device_pointer(vertices[0])), int(vertices[1])
    else:
        vertices = np.ascontiguousarray(vertices, dtype=np.float32)
        if vertices.ndim == 0 or vertices.size < 3 * stride:
            raise ValueError("vertices: an array of at least one triangle's floats is required")
        v_ptr, v_floats = _ptr(vertices), vertices.size
    if device_pointer(indices) is not None:
        i_ptr = C.c_void_p(device_pointer(indices))
    else:
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        if indices.ndim == 0 or indices.size < 3 * prim_count:       # C reads 3 * prim_count words from this array
            raise ValueError(f"indices: {indices.size} entries for {prim_count} triangles")
        i_ptr = _ptr(indices)
    n = packed_count(prim_count)
    packed = np.zeros((max(n, 1), 4), dtype=np.uint32) if want_packed else None
    ms = C.c_float(0)
    algo = {"lbvh": 0, "ploc": 1, "ploc_sah": 2, "sah": 3}[algorithm]
    _check(_lib.rts_bvh_build_device_ex(ctx.handle, v_ptr, v_floats, stride, i_ptr, prim_count,
                                        algo, radius, _ptr(packed) if want_packed else None, n if want_packed else 0,
                                        int(install), C.byref(ms)), "rts_bvh_build_device_ex")
    return (packed[:n] if want_packed else None), float(ms.value)


class ShadowContext:
    """Device-side half of the path: BVH storage buffer + the shadow dispatch."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(_lib.rts_ctx_create(device, C.byref(h)), "rts_ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            _lib.rts_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def set_bvh(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1, 4)
        _check(_lib.rts_ctx_set_bvh(self._h, _ptr(packed), packed.shape[0]), "rts_ctx_set_bvh")

    def set_option(self, key, value):
        _check(_lib.rts_ctx_set_option(self._h, key.encode(), int(value)), f"rts_ctx_set_option({key})")

    def get_option(self, key):
        v = C.c_int(0)
        _check(_lib.rts_ctx_get_option(self._h, key.encode(), C.byref(v)), f"rts_ctx_get_option({key})")
        return int(v.value)

    def trace_shadow_mask(self, constants, positions, width, height, light=None, row_begin=0, row_end=None,
                          out=None):
        """Host-pointer dispatch; returns the ``uint8[H, W]`` mask (1 = lit)."""
        positions = np.ascontiguousarray(positions, dtype=np.float32)
        if positions.size != width * height * 4:
            raise RtsError(1, "trace_shadow_mask: positions must be W*H*4 floats")
        row_end = height if row_end is None else row_end
        mask = out if out is not None else np.zeros((height, width), dtype=np.uint8)
        lp = C.byref(light) if light is not None else None
        _check(_lib.rts_trace_shadow_mask(self._h, C.byref(constants), lp, _ptr(positions), width, height,
                                          row_begin, row_end, _ptr(mask)), "rts_trace_shadow_mask")
        return mask

    def trace_shadow_mask_device(self, constants, d_positions, width, height, d_mask, light=None, row_begin=0,
                                 row_end=None, stream=None):
        row_end = height if row_end is None else row_end
        lp = C.byref(light) if light is not None else None
        _check(_lib.rts_trace_shadow_mask_device(self._h, C.byref(constants), lp, C.c_void_p(d_positions), width,
                                                 height, row_begin, row_end, C.c_void_p(d_mask),
                                                 C.c_void_p(stream or 0)), "rts_trace_shadow_mask_device")

    def trace_shadow_mask_stripes_device(self, constants, d_positions, width, height, d_mask, band_rows, n_stripes,
                                         stripe, light=None, stream=None):
        """One dispatch over the interleaved bands `stripe, stripe + n_stripes, ...` of band_rows rows each."""
        lp = C.byref(light) if light is not None else None
        _check(_lib.rts_trace_shadow_mask_stripes_device(self._h, C.byref(constants), lp, C.c_void_p(d_positions),
                                                         width, height, band_rows, n_stripes, stripe,
                                                         C.c_void_p(d_mask), C.c_void_p(stream or 0)),
               "rts_trace_shadow_mask_stripes_device")

    def trace_rays(self, rays):
        """``rays``: float32[n, 8] = {o.xyz, tmax, d.xyz, 0}; returns uint8[n] (1 = not occluded)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=np.uint8)
        _check(_lib.rts_trace_rays(self._h, _ptr(rays), rays.shape[0], _ptr(out)), "rts_trace_rays")
        return out

    def trace_rays_device(self, d_rays, n, d_out, stream=None):
        """Device-pointer form of :meth:`trace_rays` (asynchronous on ``stream``)."""
        _check(_lib.rts_trace_rays_device(self._h, C.c_void_p(d_rays), n, C.c_void_p(d_out), C.c_void_p(stream or 0)),
               "rts_trace_rays_device")

    # -- plumbing ---------------------------------------------------------------------------
    def stream_create(self):
        s = C.c_void_p()
        _check(_lib.rts_stream_create(self._h, C.byref(s)), "rts_stream_create")
        return s.value

    def stream_destroy(self, stream):
        _check(_lib.rts_stream_destroy(self._h, C.c_void_p(stream)), "rts_stream_destroy")

    def malloc(self, nbytes):
        p = C.c_void_p()
        _check(_lib.rts_device_malloc(self._h, C.byref(p), nbytes), "rts_device_malloc")
        return p.value

    def free(self, ptr):
        _check(_lib.rts_device_free(self._h, C.c_void_p(ptr)), "rts_device_free")

    def h2d(self, dptr, array):
        array = np.ascontiguousarray(array)
        _check(_lib.rts_memcpy_h2d(self._h, C.c_void_p(dptr), _ptr(array), array.nbytes), "rts_memcpy_h2d")

    def d2h(self, array, dptr):
        _check(_lib.rts_memcpy_d2h(self._h, _ptr(array), C.c_void_p(dptr), array.nbytes), "rts_memcpy_d2h")

    def mem_info(self):
        """(free, total) device memory in bytes."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        _check(_lib.rts_device_mem_info(self._h, C.byref(f), C.byref(t)), "rts_device_mem_info")
        return f.value, t.value

    def synchronize(self, stream=None):
        _check(_lib.rts_stream_synchronize(self._h, C.c_void_p(stream or 0)), "rts_stream_synchronize")

    def timer_begin(self, stream=None):
        _check(_lib.rts_timer_begin(self._h, C.c_void_p(stream or 0)), "rts_timer_begin")

    def timer_end(self, stream=None):
        _check(_lib.rts_timer_end(self._h, C.c_void_p(stream or 0)), "rts_timer_end")

    def timer_elapsed_ms(self):
        ms = C.c_float(0)
        _check(_lib.rts_timer_elapsed_ms(self._h, C.byref(ms)), "rts_timer_elapsed_ms")
        return float(ms.value)

    def timer_mark(self, slot, stream=None):
        _check(_lib.rts_timer_mark(self._h, C.c_void_p(stream or 0), slot), "rts_timer_mark")

    def timer_between_ms(self, slot_a, slot_b):
        ms = C.c_float(0)
        _check(_lib.rts_timer_between_ms(self._h, slot_a, slot_b, C.byref(ms)), "rts_timer_between_ms")
        return float(ms.value)

    def last_kernel_name(self):
        return _lib.rts_ctx_last_kernel_name(self._h).decode()

    def set_tile_order(self, order):
        if order is None:
            _check(_lib.rts_ctx_set_tile_order(self._h, None, 0), "rts_ctx_set_tile_order")
            return
        order = np.ascontiguousarray(order, np.uint32)
        _check(_lib.rts_ctx_set_tile_order(self._h, _ptr(order), order.size), "rts_ctx_set_tile_order")

    def read_wave_stats(self, waves):
        out = np.zeros((waves, 4), dtype=np.uint64)
        _check(_lib.rts_ctx_read_wave_stats(self._h, _ptr(out), waves), "rts_ctx_read_wave_stats")
        return out

    def read_wave_realtime(self, waves):
        out = np.zeros((waves, 4), dtype=np.uint64)
        _check(_lib.rts_ctx_read_wave_realtime(self._h, _ptr(out), waves), "rts_ctx_read_wave_realtime")
        return out

    def autotune(self, constants, d_positions, width, height, d_mask, light=None, stripes=None):
        """rts_ctx_autotune(_stripes): times the candidate kernels, launch options and split tables on this dispatch (stripes =
        (band_rows, n_stripes, stripe) for one rank's interleaved stripe), keeps the fastest; returns (kernel id, ms)."""
        chosen, ms = C.c_int(-1), C.c_float(0)
        lp = C.byref(light) if light is not None else None
        if stripes is not None:
            _check(_lib.rts_ctx_autotune_stripes(self._h, C.byref(constants), lp, C.c_void_p(d_positions), width, height, stripes[0],
                                                 stripes[1], stripes[2], C.c_void_p(d_mask), C.byref(chosen), C.byref(ms)),
                   "rts_ctx_autotune_stripes")
        else:
            _check(_lib.rts_ctx_autotune(self._h, C.byref(constants), lp, C.c_void_p(d_positions), width, height,
                                         C.c_void_p(d_mask), C.byref(chosen), C.byref(ms)), "rts_ctx_autotune")
        return int(chosen.value), float(ms.value)

    def plan_tile_order(self, constants, d_positions, width, height, d_mask, light=None, stripes=None, xcd_square=32, life_block=0):
        """rts_ctx_plan_tile_order: measures this dispatch, installs the longest-first, XCD-dealt tile order; returns the tiles ordered."""
        tiles = C.c_uint32(0)
        band, n, r = stripes if stripes is not None else (0, 1, 0)
        _check(_lib.rts_ctx_plan_tile_order(self._h, C.byref(constants), C.byref(light) if light is not None else None, C.c_void_p(d_positions),
                                            width, height, band, n, r, C.c_void_p(d_mask), xcd_square, life_block, C.byref(tiles)),
               "rts_ctx_plan_tile_order")
        return int(tiles.value)

    def split_plan(self):
        """Parameters of the installed split table as a dict (None without a table): rts_ctx_get_split_plan."""
        plan = SplitPlan()
        if _lib.rts_ctx_get_split_plan(self._h, C.byref(plan)) != 0:
            return None
        return {"min_life_us": float(plan.min_life_us), "end_after_us": float(plan.end_after_us), "piece_us": float(plan.piece_us),
                "front_life_us": float(plan.front_life_us), "front_share": float(plan.front_share), "max_pieces": int(plan.max_pieces),
                "max_tiles": int(plan.max_tiles), "xcd_square": int(plan.xcd_square), "life_block": int(plan.life_block)}

    def plan_splits(self, constants, d_positions, width, height, d_mask, light=None, min_life_us=20.0, piece_us=10.0,
                    max_pieces=8, max_tiles=0, row_begin=0, row_end=None, stripes=None, prev=None, end_after_us=0.0, front_life_us=0.0, front_share=0.0,
                    xcd_square=0, life_block=0):
        """rts_ctx_plan_splits(_stripes): measures the dispatch, installs the split table; returns (tiles, pieces).
        stripes = (band_rows, n_stripes, stripe) plans the interleaved-stripe dispatch; prev = (stats, realtime) arrays of an
        earlier frame (read_wave_stats / read_wave_realtime) instead of a measuring launch."""
        plan = SplitPlan(min_life_us, end_after_us, piece_us, front_life_us, front_share, max_pieces, max_tiles, xcd_square, life_block, 0, None, None, 0)
        keep = None
        if prev is not None:
            keep = (np.ascontiguousarray(prev[0], np.uint64), np.ascontiguousarray(prev[1], np.uint64))
            plan.prev_stats, plan.prev_realtime, plan.prev_waves = keep[0].ctypes.data, keep[1].ctypes.data, keep[0].size // 4
        tiles, pieces = C.c_uint32(0), C.c_uint32(0)
        lp = C.byref(light) if light is not None else None
        if stripes is not None:
            _check(_lib.rts_ctx_plan_splits_stripes(self._h, C.byref(constants), lp, C.c_void_p(d_positions), width, height,
                                                    stripes[0], stripes[1], stripes[2], C.c_void_p(d_mask), C.byref(plan),
                                                    C.byref(tiles), C.byref(pieces)), "rts_ctx_plan_splits_stripes")
        else:
            row_end = height if row_end is None else row_end
            _check(_lib.rts_ctx_plan_splits(self._h, C.byref(constants), lp, C.c_void_p(d_positions), width, height, row_begin,
                                            row_end, C.c_void_p(d_mask), C.byref(plan), C.byref(tiles), C.byref(pieces)),
                   "rts_ctx_plan_splits")
        return int(tiles.value), int(pieces.value)

    def read_piece_stats(self, pieces, clocks=True):
        """(records uint32[pieces, 8], clocks uint64[pieces, 8] or None): rts_ctx_read_piece_stats."""
        rec = np.zeros((pieces, 8), np.uint32)
        clk = np.zeros((pieces, 8), np.uint64) if clocks else None
        _check(_lib.rts_ctx_read_piece_stats(self._h, _ptr(rec), _ptr(clk) if clocks else None, pieces), "rts_ctx_read_piece_stats")
        return rec, clk

    def selftest_reciprocal(self):
        """(patterns checked, patterns that differ from the IEEE division, an example): rts_selftest_reciprocal."""
        out = np.zeros(3, np.uint64)
        _check(_lib.rts_selftest_reciprocal(self._h, _ptr(out)), "rts_selftest_reciprocal")
        return int(out[0]), int(out[1]), int(out[2])

    def clear_splits(self):
        _check(_lib.rts_ctx_clear_splits(self._h), "rts_ctx_clear_splits")

    def clock_probe_mhz(self, rows):
        """Shader clock held during the launches since set_option("clock_probe", rows) (the timed launches themselves)."""
        out = np.zeros((rows, 4), np.uint64)
        _check(_lib.rts_ctx_read_clock_probe(self._h, _ptr(out), rows), "rts_ctx_read_clock_probe")
        self.last_clock_probe = out.copy()                      # (tools: wave lifetimes, hardware slots)
        end = out[:, 1] & np.uint64(0x0000FFFFFFFFFFFF)         # the top 16 bits carry the wave's hardware slot (HW_ID)
        ok = (end > out[:, 0]) & (out[:, 3] > out[:, 2])
        if not ok.any():
            return None
        return float((end[ok] - out[ok, 0]).astype(np.float64).sum() / (out[ok, 3] - out[ok, 2]).astype(np.float64).sum() * 100.0)

    def measure_shader_clock_mhz(self, trace, waves, launches=8):
        """Clock the chip holds while `trace()` (one dispatch of a packet kernel with `waves` one-wave workgroups) runs
        back to back: shader clocks per 100 MHz realtime tick, summed over every wave of the last launch
        (MI355X_MICROARCH.md, DVFS give-back item 6).  Diagnostics build of the kernel; results are not touched."""
        self.set_option("wave_stats", waves)
        try:
            for _ in range(launches):
                trace()
            self.synchronize()
            st = self.read_wave_stats(waves)
            rt = self.read_wave_realtime(waves)
        finally:
            self.set_option("wave_stats", 0)
        ok = (rt[:, 1] > rt[:, 0]) & (st[:, 1] > st[:, 0])
        if not ok.any():
            return None
        clocks = (st[ok, 1] - st[ok, 0]).astype(np.float64).sum()
        ticks = (rt[ok, 1] - rt[ok, 0]).astype(np.float64).sum()
        return float(clocks / ticks * 100.0)


# -- harness entry points -----------------------------------------------------------------------
def primary_positions(packed, eye, target, fovy, width, height, threads=0):
    """G-buffer position target (camera-relative closest hit per pixel); see include/rts_scene.h."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1, 4)
    pos = np.zeros((height, width, 4), dtype=np.float32)
    e = (C.c_float * 3)(*[float(x) for x in eye])
    t = (C.c_float * 3)(*[float(x) for x in target])
    hits = C.c_uint64(0)
    _check(_lib.rtsh_primary_positions(_ptr(packed), packed.shape[0], e, t, fovy, width, height, _ptr(pos),
                                       C.byref(hits), threads), "rtsh_primary_positions")
    return pos, int(hits.value)


def primary_gbuffer(packed, eye, target, fovy, width, height, threads=0):
    """Positions + normals targets (host)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1, 4)
    pos = np.zeros((height, width, 4), dtype=np.float32)
    nrm = np.zeros((height, width, 4), dtype=np.float32)
    e = (C.c_float * 3)(*[float(x) for x in eye])
    t = (C.c_float * 3)(*[float(x) for x in target])
    hits = C.c_uint64(0)
    _check(_lib.rtsh_primary_gbuffer(_ptr(packed), packed.shape[0], e, t, fovy, width, height, _ptr(pos), _ptr(nrm),
                                     C.byref(hits), threads), "rtsh_primary_gbuffer")
    return pos, nrm, int(hits.value)


def primary_gbuffer_device(ctx, eye, target, fovy, width, height, d_positions, d_normals=None, stream=None):
    """The G-buffer pass on the GPU, through the BVH uploaded to `ctx` (device pointers, asynchronous)."""
    e = (C.c_float * 3)(*[float(x) for x in eye])
    t = (C.c_float * 3)(*[float(x) for x in target])
    _check(_lib.rtsh_primary_gbuffer_device(ctx.handle, e, t, fovy, width, height, C.c_void_p(d_positions),
                                            C.c_void_p(d_normals or 0), C.c_void_p(stream or 0)),
           "rtsh_primary_gbuffer_device")


def combine(constants, light, positions, normals, mask):
    """Combine.frag on the host: uint8[H, W, 3]."""
    H, W = mask.shape
    normals = np.ascontiguousarray(normals, np.float32)
    positions = np.ascontiguousarray(positions, np.float32) if positions is not None else None
    mask = np.ascontiguousarray(mask, np.uint8)
    rgb = np.zeros((H, W, 3), np.uint8)
    lp = C.byref(light) if light is not None else None
    _check(_lib.rtsh_combine(C.byref(constants), lp, _ptr(positions) if positions is not None else None, _ptr(normals),
                             _ptr(mask), W, H, _ptr(rgb)), "rtsh_combine")
    return rgb


def combine_device(ctx, constants, light, d_positions, d_normals, d_mask, width, height, d_rgb, stream=None):
    """Combine.frag on the GPU: device pointers, d_rgb = width*height*3 bytes, asynchronous."""
    lp = C.byref(light) if light is not None else None
    _check(_lib.rtsh_combine_device(ctx.handle, C.byref(constants), lp, C.c_void_p(d_positions or 0), C.c_void_p(d_normals),
                                    C.c_void_p(d_mask), width, height, C.c_void_p(d_rgb), C.c_void_p(stream or 0)),
           "rtsh_combine_device")


def write_ppm(path, rgb):
    """Binary PPM (P6) image dump."""
    H, W, _ = rgb.shape
    with open(path, "wb") as fh:
        fh.write(f"P6\n{W} {H}\n255\n".encode())
        fh.write(np.ascontiguousarray(rgb, np.uint8).tobytes())
    return path


_BLOB_MAGIC = b"RTSBVH01"


def save_bvh(path, packed):
    """Serialises a packed node stream (cacheable scenes): magic, vec4 count, raw little-endian bytes."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1, 4)
    bvh_validate(packed)
    with open(path, "wb") as fh:
        fh.write(_BLOB_MAGIC)
        fh.write(np.array([packed.shape[0]], np.uint64).tobytes())
        fh.write(packed.tobytes())
    return path


def load_bvh(path):
    with open(path, "rb") as fh:
        if fh.read(8) != _BLOB_MAGIC:
            raise RtsError(5, f"load_bvh: {path} is not a packed-BVH blob")
        n = int(np.frombuffer(fh.read(8), np.uint64)[0])
        packed = np.frombuffer(fh.read(n * 16), np.uint32).reshape(-1, 4).copy()
    if packed.shape[0] != n:
        raise RtsError(5, f"load_bvh: {path} is truncated")
    bvh_validate(packed)
    return packed


def obj_load(path):
    """OBJ -> flat ``float32[3T, 8]`` Vertex stream + ``indices[i] = i`` (loadModel semantics)."""
    n = C.c_uint32(0)
    _check(_lib.rtsh_obj_load(path.encode(), None, 0, C.byref(n), None, None), "rtsh_obj_load")
    verts = np.zeros((max(n.value, 1), 8), dtype=np.float32)
    lo = (C.c_float * 3)()
    hi = (C.c_float * 3)()
    _check(_lib.rtsh_obj_load(path.encode(), _ptr(verts), verts.shape[0], C.byref(n), lo, hi), "rtsh_obj_load")
    verts = verts[:n.value]
    return verts, np.arange(n.value, dtype=np.uint32), np.array(lo[:], np.float32), np.array(hi[:], np.float32)


def obj_parse_float(text):
    used = C.c_int(0)
    v = _lib.rtsh_obj_parse_float(text.encode(), C.byref(used))
    return np.float32(v), int(used.value)
