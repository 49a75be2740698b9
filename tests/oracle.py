"""ctypes view of oracle/librts_oracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_path = os.path.join(ROOT, "oracle", "librts_oracle.so")
if not os.path.exists(_path):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
_o = C.CDLL(_path)

_o.orc_packed_count.restype = C.c_uint64
_o.orc_packed_count.argtypes = [C.c_uint32]
_o.orc_bvh_build.restype = C.c_int
_o.orc_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
_o.orc_bvh_build_ties_by_prim.restype = C.c_int
_o.orc_bvh_build_ties_by_prim.argtypes = _o.orc_bvh_build.argtypes
_o.orc_any_hit.restype = C.c_int
_o.orc_any_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
_o.orc_trace_rays.restype = None
_o.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
_o.orc_shadow_mask.restype = None
_o.orc_shadow_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
_o.orc_gen_rays.restype = None
_o.orc_gen_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
_o.orc_brute_force_rays.restype = None
_o.orc_brute_force_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
_o.orc_ray_box.restype = C.c_int
_o.orc_ray_box.argtypes = [C.c_void_p] * 4
_o.orc_ray_tri.restype = C.c_int
_o.orc_ray_tri.argtypes = [C.c_void_p] * 5
_o.orc_epsilon_for.restype = C.c_float
_o.orc_epsilon_for.argtypes = [C.c_float, C.c_uint32]
_o.orc_max_threads.restype = C.c_int
_o.orc_primary_gbuffer.restype = None
_o.orc_primary_gbuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]


class OLight(C.Structure):
    """Same layout as rts_light (include/rts.h)."""
    _fields_ = [("type", C.c_uint32), ("nsamples", C.c_uint32), ("xyz", C.c_float * 3), ("table", C.c_uint32),
                ("offsets", (C.c_float * 4) * 64)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_light(kind, xyz, offsets=None, nsamples=None):
    """offsets: the sample offsets; nsamples < len(offsets) = per-pixel jitter over the whole table (rts_light.table)."""
    lt = OLight()
    lt.type = kind
    lt.nsamples = 1
    for i in range(3):
        lt.xyz[i] = np.float32(xyz[i])
    if offsets is not None:
        offsets = np.asarray(offsets, np.float32)
        lt.nsamples = offsets.shape[0]
        if nsamples is not None and nsamples != offsets.shape[0]:
            lt.nsamples, lt.table = nsamples, offsets.shape[0]
        for j in range(offsets.shape[0]):
            for i in range(3):
                lt.offsets[j][i] = offsets[j, i]
    return lt


def light_from_product(light, constants):
    """Oracle light equivalent to what the product does with (light | None, constants)."""
    if light is None:
        return make_light(0, [constants.lightDirection[i] for i in range(3)])
    lt = OLight()
    C.memmove(C.byref(lt), C.byref(light), C.sizeof(OLight))
    if lt.nsamples == 0:
        lt.nsamples = 1
    return lt


def max_threads():
    return int(_o.orc_max_threads())


def bvh_build(vertices, stride, indices, prim_count, sah_limit=1000000, want_nodes=False, ties_by_prim=False):
    """ties_by_prim: equal centroids ordered by triangle id (the rule of the GPU SAH builder) instead of std::sort's."""
    vertices = np.ascontiguousarray(vertices, np.float32)
    indices = np.ascontiguousarray(indices, np.uint32)
    n = int(_o.orc_packed_count(prim_count))
    packed = np.zeros((n, 4), np.uint32)
    nodes = np.zeros((2 * prim_count - 1, 8), np.uint32) if want_nodes else None
    fn = _o.orc_bvh_build_ties_by_prim if ties_by_prim else _o.orc_bvh_build
    st = fn(_p(vertices), stride, _p(indices), prim_count, sah_limit, _p(packed), _p(nodes) if want_nodes else None)
    if st != 0:
        raise ValueError("oracle builder rejected the input")
    return (packed, nodes) if want_nodes else packed


def any_hit(packed, o4, d4):
    packed = np.ascontiguousarray(packed, np.uint32)
    o4 = np.ascontiguousarray(o4, np.float32)
    d4 = np.ascontiguousarray(d4, np.float32)
    v, l = C.c_uint32(0), C.c_uint32(0)
    hit = _o.orc_any_hit(_p(packed), _p(o4), _p(d4), C.byref(v), C.byref(l))
    return bool(hit), int(v.value), int(l.value)


def trace_rays(packed, rays, threads=0):
    packed = np.ascontiguousarray(packed, np.uint32)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros(rays.shape[0], np.uint8)
    sums = np.zeros(2, np.uint64)
    _o.orc_trace_rays(_p(packed), _p(rays), rays.shape[0], _p(out), _p(sums), threads)
    return out, int(sums[0]), int(sums[1])


def shadow_mask(packed, constants_array, light, positions, W, H, row_begin=0, row_end=None, threads=0,
                per_ray=False, out=None):
    packed = np.ascontiguousarray(packed, np.uint32)
    k = np.ascontiguousarray(constants_array, np.float32)
    positions = np.ascontiguousarray(positions, np.float32)
    row_end = H if row_end is None else row_end
    mask = out if out is not None else np.zeros((H, W), np.uint8)
    sums = np.zeros(2, np.uint64)
    pv = np.zeros((H, W), np.uint32) if per_ray else None
    pl = np.zeros((H, W), np.uint32) if per_ray else None
    _o.orc_shadow_mask(_p(packed), _p(k), C.byref(light), _p(positions), W, H, row_begin, row_end, _p(mask),
                       _p(sums), _p(pv) if per_ray else None, _p(pl) if per_ray else None, threads)
    if per_ray:
        return mask, int(sums[0]), int(sums[1]), pv, pl
    return mask, int(sums[0]), int(sums[1])


def gen_rays(constants_array, light, positions):
    k = np.ascontiguousarray(constants_array, np.float32)
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 4)
    ns = max(1, light.nsamples)
    rays = np.zeros((positions.shape[0] * ns, 8), np.float32)
    _o.orc_gen_rays(_p(k), C.byref(light), _p(positions), positions.shape[0], _p(rays))
    return rays


def brute_force_rays(packed, prim_count, rays):
    packed = np.ascontiguousarray(packed, np.uint32)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros(rays.shape[0], np.uint8)
    _o.orc_brute_force_rays(_p(packed), prim_count, _p(rays), rays.shape[0], _p(out))
    return out


def primary_gbuffer(packed, eye, target, fovy, W, H, cull=True, threads=0):
    """(positions[H,W,4], normals[H,W,4], hits): the G-buffer targets of Model.frag:35-39 by closest hit (see
    orc_primary_gbuffer); cull=False = brute force over every triangle."""
    packed = np.ascontiguousarray(packed, np.uint32)
    eye = np.ascontiguousarray(eye, np.float32)
    target = np.ascontiguousarray(target, np.float32)
    pos = np.zeros((H, W, 4), np.float32)
    nrm = np.zeros((H, W, 4), np.float32)
    hits = np.zeros(1, np.uint64)
    _o.orc_primary_gbuffer(_p(packed), _p(eye), _p(target), C.c_float(fovy), W, H, int(bool(cull)), _p(pos), _p(nrm),
                           _p(hits), threads)
    return pos, nrm, int(hits[0])


def ray_box(o3, invdir3, pmin3, pmax3):
    a = [np.ascontiguousarray(x, np.float32) for x in (o3, invdir3, pmin3, pmax3)]
    return bool(_o.orc_ray_box(*[_p(x) for x in a]))


def ray_tri(o4, d3, v0, e0, e1):
    a = [np.ascontiguousarray(x, np.float32) for x in (o4, d3, v0, e0, e1)]
    return bool(_o.orc_ray_tri(*[_p(x) for x in a]))


def epsilon_for(f, diff=13):
    return np.float32(_o.orc_epsilon_for(np.float32(f), diff))


_o.orc_combine.restype = None
_o.orc_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]


def combine(constants_array, light, positions, normals, mask):
    """Combine.frag:18-37 restated (orc_combine): uint8[H, W, 3].  light: an OLight, or None for the reference's
    directional light taken from the constants."""
    H, W = mask.shape
    k = np.ascontiguousarray(constants_array, np.float32)
    positions = np.ascontiguousarray(positions, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    rgb = np.zeros((H, W, 3), np.uint8)
    _o.orc_combine(_p(k), C.byref(light) if light is not None else None, _p(positions), _p(normals), _p(mask), W, H, _p(rgb))
    return rgb


def order_experiment(packed, constants_array, light, positions, W, H, mode):
    """Visit statistics under another child order (experiment, see orc_order_experiment): dict of per-ray / per-tile means."""
    packed = np.ascontiguousarray(packed, np.uint32)
    k = np.ascontiguousarray(constants_array, np.float32)
    positions = np.ascontiguousarray(positions, np.float32)
    out = np.zeros(8, np.uint64)
    _o.orc_order_experiment(_p(packed), _p(k), C.byref(light), _p(positions), W, H, mode, _p(out))
    tiles = int(out[0])
    return {"tiles": tiles, "union_per_tile": int(out[1]) / tiles, "longest_per_tile": int(out[2]) / tiles,
            "visits_per_ray": int(out[3]) / (tiles * 64), "occluded": int(out[4])}
