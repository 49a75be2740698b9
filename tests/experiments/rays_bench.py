"""Experiment (GPU box): generic-ray entry (rts_trace_rays_device) per kernel variant, on coherent rays (the shadow rays
of a frame, in raster order) and on incoherent ones (the same rays shuffled; random segments through the scene).
Checked against the oracle.  Lives under tests/ because it uses the oracle as its checker."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracedshadows_amd import api, workloads
import oracle

for scene, W, H in (("city", 1920, 1080), ("atrium", 1280, 720)):
    wl = workloads.prepare(scene, W, H)
    shadow = oracle.gen_rays(wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions)
    rs = np.random.RandomState(1)
    lo, hi = wl.scene.bbox_min, wl.scene.bbox_max
    n = shadow.shape[0]
    a = (lo + (hi - lo) * rs.rand(n, 3)).astype(np.float32)
    b = (lo + (hi - lo) * rs.rand(n, 3)).astype(np.float32)
    rnd = np.zeros((n, 8), np.float32); rnd[:, :3] = a; rnd[:, 3] = 1.0; rnd[:, 4:7] = b - a
    sets = {"shadow rays, raster order": shadow, "shadow rays, shuffled": shadow[rs.permutation(n)], "random segments": rnd}
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_rays, d_out = ctx.malloc(n * 32), ctx.malloc(n)
        for tag, rays in sets.items():
            want = oracle.trace_rays(wl.packed, rays)
            want = want[0] if isinstance(want, tuple) else want
            ctx.h2d(d_rays, np.ascontiguousarray(rays))
            for variant in (0, 1, 2, 3, 7):
                ctx.set_option("kernel", variant)
                ts = []
                for i in range(25):
                    ctx.timer_begin(); ctx.trace_rays_device(d_rays, n, d_out); ctx.timer_end()
                    if i >= 5: ts.append(ctx.timer_elapsed_ms())
                got = np.zeros(n, np.uint8); ctx.d2h(got, d_out)
                print(f"[{scene} {W}x{H}] {tag}: variant {variant}: {np.median(ts):.4f} ms = {n / np.median(ts) / 1e6:.2f} Grays/s; "
                      f"mismatch {(got != want.reshape(-1)).sum()}", flush=True)
        ctx.set_option("kernel", -1)
        ctx.free(d_rays); ctx.free(d_out)
