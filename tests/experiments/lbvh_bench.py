"""GPU box: build time and traversal cost of the GPU-built (LBVH) stream vs the host-built SAH stream."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracedshadows_amd import api, workloads
import oracle

for cfg in ("atrium_1080p", "city_4k"):
    wl = workloads.prepare_config(cfg)
    W, H = wl.W, wl.H
    lt = oracle.light_from_product(wl.light, wl.constants)
    with api.ShadowContext(0) as ctx:
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        api.bvh_build_device(ctx, wl.vertices, 8, wl.indices, wl.prim_count, want_packed=False)   # warm-up (hipcub, allocs)
        t0 = time.time()
        packed, ms = api.bvh_build_device(ctx, wl.vertices, 8, wl.indices, wl.prim_count, install=True)
        wall = time.time() - t0
        for name, blob in (("LBVH (GPU build)", None), ("SAH (host build)", wl.packed)):
            if blob is not None:
                ctx.set_bvh(blob)
            ref = packed if blob is None else blob
            want, V, L = oracle.shadow_mask(ref, wl.constants.as_array(), lt, wl.positions, W, H)
            got = np.zeros((H, W), np.uint8)
            ts = []
            for i in range(25):
                ctx.timer_begin(); ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light); ctx.timer_end()
                ts.append(ctx.timer_elapsed_ms())
            ctx.d2h(got, d_mask)
            print(f"[{cfg}] {name}: trace {np.median(ts[5:]):.4f} ms, mismatches vs oracle on the same stream {int((got != want).sum())}, "
                  f"nodes/ray {V / want.size:.1f}, tris/ray {L / want.size:.2f}", flush=True)
        print(f"[{cfg}] GPU build: {ms:.2f} ms on the device ({wall * 1e3:.1f} ms wall incl. H2D of {wl.vertices.nbytes / 1e6:.0f} MB vertices + D2H of the stream); "
              f"host SAH build: {wl.build_seconds * 1e3:.0f} ms", flush=True)
