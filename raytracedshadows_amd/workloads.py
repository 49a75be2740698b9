"""Harness: turns a BASELINE.json config into the inputs of the shadow dispatch.

scene generator -> real .obj text -> OBJ reader (loadModel semantics) -> BVHBuilder.build ->
camera -> G-buffer positions -> RayTracingConstants + light.  Nothing here is timed as part of the
hot path; both the GPU kernels and the CPU oracle consume the arrays this produces.
"""
import os
import time

import numpy as np

from . import api, scenes

#: BASELINE.json "configs", in order.  (scene, W, H, light kind, samples per pixel)
CONFIGS = {
    "cornell_256": ("cornell", 256, 256, "point", 1),         # configs[0]
    "atrium_1080p": ("atrium", 1920, 1080, "point", 1),       # configs[1]
    "city_4k": ("city", 3840, 2160, "point", 1),              # configs[2]  (headline; [3] = same, striped)
    "city_4k_soft16": ("city", 3840, 2160, "point", 16),      # configs[4]
    "courtyard_4k": ("courtyard", 3840, 2160, "point", 1),    # configs[2] on the San-Miguel-class stand-in (hard case)
    "courtyard_4k_soft16": ("courtyard", 3840, 2160, "point", 16),
    "city_4k_soft16_wide": ("city", 3840, 2160, "point", 16),  # configs[4] with a five times larger light (SOFT_RADIUS)
    "city_4k_directional": ("city", 3840, 2160, "directional", 1),   # the reference's own light (RayTracedShadows.cpp:245, comp:128-151) at the headline size
    "city_4k_soft16_pp": ("city", 3840, 2160, "point", 16),    # configs[4] as a ray-packet STRESS: per-pixel jitter (PER_PIXEL_TABLE)
    "courtyard_4k_soft16_pp": ("courtyard", 3840, 2160, "point", 16),
    "calib_4k": ("calib", 3840, 2160, "point", 1),            # counter calibration only (see scenes.calib)
}


#: radius of the jittered light as a fraction of the scene's diagonal (default 1 %)
SOFT_RADIUS = {"city_4k_soft16_wide": 0.05}

#: configs whose light is a table of this many offsets from which every pixel takes `spp` entries, starting at a position
#: hashed from the pixel index (rts_light.table): neighbouring pixels aim at different points of the light in the same pass
PER_PIXEL_TABLE = {"city_4k_soft16_pp": 64, "courtyard_4k_soft16_pp": 64}


class Workload:
    pass


def prepare(scene_name, W, H, light="point", spp=1, via_obj=True, threads=0, log=None, packed=None, radius=0.01, table=0):
    say = log or (lambda *a: None)
    wl = Workload()
    t0 = time.time()
    sc = scenes.SCENES[scene_name]() if isinstance(scene_name, str) else scene_name
    say(f"scene {sc.name}: {sc.triangle_count} triangles generated in {time.time() - t0:.2f}s")
    if via_obj:
        t0 = time.time()
        path = os.path.join(scenes.cache_dir(), f"{sc.name}_{os.getpid()}.obj")   # one file per process (ranks)
        sc.write_obj(path)
        verts, indices, lo, hi = api.obj_load(path)
        say(f"obj written + parsed in {time.time() - t0:.2f}s ({os.path.getsize(path) / 1e6:.1f} MB)")
        os.remove(path)
    else:
        verts, indices = sc.flat()
    prim_count = verts.shape[0] // 3
    t0 = time.time()
    if packed is None:
        builder = api.BVHBuilder(threads=threads).build(verts, 8, indices, prim_count)
        packed_nodes, nodes = builder.m_packedNodes, builder.m_nodes
        wl.build_seconds = time.time() - t0
        say(f"BVH built in {wl.build_seconds:.2f}s ({packed_nodes.nbytes / 1e6:.1f} MB packed)")
    else:                                        # a stream built elsewhere (e.g. broadcast from rank 0)
        packed_nodes, nodes = np.ascontiguousarray(packed, np.uint32).reshape(-1, 4), None
        assert api.bvh_validate(packed_nodes) == prim_count
        wl.build_seconds = 0.0
    t0 = time.time()
    positions, hits = api.primary_positions(packed_nodes, sc.eye, sc.target, sc.fovy, W, H, threads)
    say(f"G-buffer positions {W}x{H} in {time.time() - t0:.2f}s ({hits / (W * H) * 100:.1f}% of pixels hit geometry)")

    wl.scene = sc
    wl.W, wl.H, wl.spp = W, H, spp
    wl.prim_count = prim_count
    wl.vertices, wl.indices = verts, indices
    wl.packed = packed_nodes
    wl.nodes = nodes
    wl.positions = positions
    wl.constants = api.RayTracingConstants.make(sc.eye, sc.light_direction, W, H, sc.target - sc.eye)
    return relight(wl, light, spp, radius, table)


def relight(wl, light="point", spp=1, radius=0.01, table=0):
    """The same scene, camera and G-buffer under another light / sample count (a shallow copy of `wl`)."""
    import copy
    wl = copy.copy(wl)
    sc = wl.scene
    wl.spp = spp
    if light == "directional":
        wl.light = None if spp <= 1 else api.Light.make(api.Light.DIRECTIONAL, sc.light_direction,
                                                        scenes.jitter_offsets(spp, 0.05))
    else:
        r = radius * float(np.linalg.norm(sc.bbox_max - sc.bbox_min))
        wl.light = api.Light.make(api.Light.POINT, sc.light_point,
                                  scenes.jitter_offsets(max(spp, table), r) if spp > 1 else None,
                                  nsamples=spp if spp > 1 else None)
    wl.rays = wl.W * wl.H * max(1, spp)
    return wl


def prepare_config(name, cache=False, **kw):
    """Inputs of a BASELINE config.  cache=True keeps the derived arrays (flat vertices, packed BVH, G-buffer positions)
    in the scene cache directory, keyed by the config, the generator source and the library build, so that the profiler
    passes of bench.py (child processes of one run) do not rebuild them."""
    scene, W, H, light, spp = CONFIGS[name]
    radius = SOFT_RADIUS.get(name, 0.01)
    table = PER_PIXEL_TABLE.get(name, 0)
    if not cache:
        return prepare(scene, W, H, light=light, spp=spp, radius=radius, table=table, **kw)
    import hashlib
    h = hashlib.sha256()
    h.update(repr((name, scene, W, H)).encode())
    h.update(open(scenes.__file__, "rb").read())
    h.update(open(__file__, "rb").read())
    st = os.stat(api.lib_path())
    h.update(repr((st.st_size, int(st.st_mtime))).encode())
    path = os.path.join(scenes.cache_dir(), f"wl_{name}_{h.hexdigest()[:16]}.npz")
    say = kw.get("log") or (lambda *a: None)
    if os.path.exists(path):
        try:
            z = np.load(path)
            wl = Workload()
            wl.scene = scenes.SCENES[scene]()
            wl.W, wl.H = W, H
            wl.vertices, wl.indices, wl.packed, wl.positions = z["vertices"], z["indices"], z["packed"], z["positions"]
            wl.nodes = None
            wl.prim_count = wl.vertices.shape[0] // 3
            wl.build_seconds = float(z["build_seconds"])
            sc = wl.scene
            wl.constants = api.RayTracingConstants.make(sc.eye, sc.light_direction, W, H, sc.target - sc.eye)
            say(f"workload {name} loaded from {path}")
            return relight(wl, light, spp, radius, table)
        except Exception as e:                       # a torn or stale file: rebuild
            say(f"cache {path} unusable ({e!r}); rebuilding")
    wl = prepare(scene, W, H, light=light, spp=spp, radius=radius, table=table, **kw)
    tmp = f"{path}.{os.getpid()}.tmp.npz"
    np.savez(tmp, vertices=wl.vertices, indices=wl.indices, packed=wl.packed, positions=wl.positions,
             build_seconds=np.float64(wl.build_seconds))
    os.replace(tmp, path)
    return wl
