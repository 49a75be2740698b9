"""Experiment (GPU box): lifetimes and hardware slots of the probe waves (first wave of every tile row) of the EVERYDAY
instantiation -- the one that is timed and has no other stamps."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from raytracedshadows_amd import api, workloads
cfg = sys.argv[1] if len(sys.argv) > 1 else "city_4k"
wl = workloads.prepare_config(cfg, cache=True)
W, H = wl.W, wl.H
rows = (H + 7) // 8
with api.ShadowContext(0) as ctx:
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    ctx.h2d(d_pos, wl.positions)
    for k in (3, 8, 3, 8):
        ctx.set_option("kernel", k)
        for _ in range(300):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        ctx.set_option("clock_probe", rows)
        lives, slots = [], set()
        for _ in range(40):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            mhz = ctx.clock_probe_mhz(rows)
            o = ctx.last_clock_probe
            lives.append((o[:, 3] - o[:, 2]).astype(np.float64) / 100.0)
            hw = (o[:, 1] >> np.uint64(48)).astype(np.int64)
            slots |= set((hw & 0xF).tolist())
        ctx.set_option("clock_probe", 0)
        lives = np.concatenate(lives)
        print(f"{cfg} {ctx.last_kernel_name()}: probe waves (column 0 of every tile row): life mean {lives.mean():.2f} us, p50 {np.percentile(lives, 50):.2f}, "
              f"p90 {np.percentile(lives, 90):.2f}; wave slots seen per SIMD: {sorted(slots)}; clock {mhz:.0f} MHz", flush=True)
