"""GPU box: a scene that changes every frame -- the whole frame on the device with a rebuild of the BVH per frame.

Per frame: the 1M triangles move (a travelling wave displaces every vertex, computed on the host), upload, SAH build on the device
(the reference's tree), G-buffer pass, shadow mask, combine.  First and last frame: the stream is read back and mask and G-buffer
are checked against the oracle on that stream.   python tests/experiments/dynamic_scene.py [city|courtyard] [frames] [WxH]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from raytracedshadows_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "city"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
W, H = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "3840x2160").split("x")]
sc = scenes.SCENES[name]()
base, idx = sc.flat()
P = sc.triangle_count
amp = 0.002 * float(np.linalg.norm(sc.bbox_max - sc.bbox_min))
k = api.RayTracingConstants.make(sc.eye, sc.light_direction, W, H)
light = api.Light.make(api.Light.POINT, sc.light_point)
with api.ShadowContext(0) as ctx:
    d_pos, d_nrm, d_mask, d_rgb = ctx.malloc(W * H * 16), ctx.malloc(W * H * 16), ctx.malloc(W * H), ctx.malloc(W * H * 3)
    rows = []
    for f in range(frames):
        v = base.copy()
        v[:, 1] += (amp * np.sin(0.35 * base[:, 0] + 0.6 * f)).astype(np.float32)          # the geometry of this frame
        check = f in (0, frames - 1)
        t0 = time.time()
        packed, build_ms = api.bvh_build_device(ctx, v, 8, idx, P, install=True, want_packed=check)
        t1 = time.time()
        ctx.timer_mark(0)
        api.primary_gbuffer_device(ctx, sc.eye, sc.target, sc.fovy, W, H, d_pos, d_nrm)
        ctx.timer_mark(1)
        ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
        ctx.timer_mark(2)
        api.combine_device(ctx, k, light, d_pos, d_nrm, d_mask, W, H, d_rgb)
        ctx.timer_mark(3)
        ctx.synchronize()
        t2 = time.time()
        rows.append(((t1 - t0) * 1e3, build_ms, ctx.timer_between_ms(0, 1), ctx.timer_between_ms(1, 2), ctx.timer_between_ms(2, 3), (t2 - t0) * 1e3))
        if check:
            pos, mask = np.zeros((H, W, 4), np.float32), np.zeros((H, W), np.uint8)
            ctx.d2h(pos, d_pos); ctx.d2h(mask, d_mask)
            want_pos = oracle.primary_gbuffer(packed, sc.eye, sc.target, sc.fovy, W, H)[0]
            want, _, _ = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(light, k), pos, W, H)
            print(f"frame {f}: G-buffer == oracle: {bool((pos.view(np.uint32) == np.asarray(want_pos).view(np.uint32)).all())}, "
                  f"mask mismatches vs oracle on the frame's own stream: {int((mask != want).sum())}, lit {float(mask.mean()):.3f}", flush=True)
    r = np.array(rows[2:])                                              # (the first frames grow the context's buffers)
    m = np.median(r, 0)
    print(f"{name}: {P} triangles rebuilt every frame, {W}x{H}, {len(r)} frames, medians: build call {m[0]:.2f} ms (device {m[1]:.2f}), "
          f"G-buffer {m[2]:.3f}, shadow mask {m[3]:.3f}, combine {m[4]:.3f}; whole frame {m[5]:.2f} ms = {1e3 / m[5]:.0f} frames/s "
          f"(host-side vertex animation not counted)")
