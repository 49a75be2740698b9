"""GPU box: beyond the baseline's sizes -- a heightfield of 2 n^2 triangles (default n = 2236: 10 M triangles, an 800 MB stream): SAH build
on the device (median splits above the reference's 1M limit, SAH below), validation of the stream, G-buffer and a 4K shadow mask through
it, mask against the oracle on the same stream.   python tests/experiments/big_scene.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from raytracedshadows_amd import api, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
W, H = 3840, 2160
t0 = time.time()
sc = scenes.terrain(n)
P = sc.triangle_count
verts = np.ascontiguousarray(sc.verts, np.float32)
idx = np.ascontiguousarray(sc.faces.reshape(-1), np.uint32)
print(f"terrain {n} x {n}: {P} triangles, {verts.shape[0]} vertices ({time.time() - t0:.1f}s)", flush=True)
with api.ShadowContext(0) as ctx:
    t0 = time.time()
    packed, ms = api.bvh_build_device(ctx, verts, 3, idx, P, install=True)
    wall = time.time() - t0
    _, ms2 = api.bvh_build_device(ctx, verts, 3, idx, P, install=True, want_packed=False)
    print(f"SAH on the device: {ms:.1f} ms (second build {ms2:.1f} ms), {wall:.2f}s for the first call incl. the {packed.nbytes / 1e6:.0f} MB read-back; "
          f"validate: {api.bvh_validate(packed)} triangles", flush=True)
    k = api.RayTracingConstants.make(sc.eye, sc.light_direction, W, H)
    light = api.Light.make(api.Light.POINT, sc.light_point)
    d_pos, d_mask = ctx.malloc(W * H * 16), ctx.malloc(W * H)
    api.primary_gbuffer_device(ctx, sc.eye, sc.target, sc.fovy, W, H, d_pos)
    for _ in range(50):
        ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
    ts = []
    for _ in range(20):
        ctx.timer_mark(0); ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light); ctx.timer_mark(1)
        ts.append(ctx.timer_between_ms(0, 1))
    pos, mask = np.zeros((H, W, 4), np.float32), np.zeros((H, W), np.uint8)
    ctx.d2h(pos, d_pos); ctx.d2h(mask, d_mask)
    want, V, L = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(light, k), pos, W, H)
    print(f"4K shadow mask through it: {np.median(ts):.3f} ms ({W * H / np.median(ts) / 1e6:.1f} Grays/s), {V / want.size:.1f} nodes per ray, "
          f"mismatches vs oracle on the same stream: {int((mask != want).sum())}, lit {float(mask.mean()):.3f}")
