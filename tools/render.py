"""Human-checkable output: OBJ scene -> BVH -> G-buffer (GPU) -> shadow mask (GPU) -> combine -> PPM.

    python tools/render.py --config atrium_1080p --out gpurun_out/atrium.ppm [--spp 16] [--save-bvh x.bvh]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="atrium_1080p")
    ap.add_argument("--out", default="gpurun_out/render.ppm")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--save-bvh", default="")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    scene, W, H, light, spp = workloads.CONFIGS[args.config]
    wl = workloads.prepare(scene, W, H, light=light, spp=args.spp or spp, log=print)
    if args.save_bvh:
        api.save_bvh(args.save_bvh, wl.packed)
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_nrm, d_mask = ctx.malloc(W * H * 16), ctx.malloc(W * H * 16), ctx.malloc(W * H)
        t0 = time.time()
        api.primary_gbuffer_device(ctx, wl.scene.eye, wl.scene.target, wl.scene.fovy, W, H, d_pos, d_nrm)
        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        print(f"G-buffer + shadow mask on the GPU: {(time.time() - t0) * 1e3:.2f} ms (first call, incl. launch)")
        d_rgb = ctx.malloc(W * H * 3)
        api.combine_device(ctx, wl.constants, wl.light, d_pos, d_nrm, d_mask, W, H, d_rgb)     # the frame stays on the device
        ctx.synchronize()
        rgb, mask = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.uint8)
        ctx.d2h(rgb, d_rgb); ctx.d2h(mask, d_mask)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    api.write_ppm(args.out, rgb)
    print(f"wrote {args.out}: {W}x{H}, lit fraction {float((mask > 0).mean()):.3f}")


if __name__ == "__main__":
    main()
