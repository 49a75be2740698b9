"""Experiment (GPU box): option "tiles_per_wave" of the everyday launches -- a wave walks 1..4 tile columns of its tile row one
after the other, the next tile's texel requested before the current walk -- timed per kernel, every mask compared with the
oracle's.   python tests/experiments/tiles_per_wave.py city_4k [courtyard_4k ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracedshadows_amd import api, workloads
import oracle as orc

TPW = [int(v) for v in os.environ.get("TPW", "1,2,3,4").split(",")]
for cfg in sys.argv[1:] or ["city_4k"]:
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    expect = orc.shadow_mask(wl.packed, wl.constants.as_array(), orc.light_from_product(wl.light, wl.constants), wl.positions, W, H)[0]
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_m = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for rep in range(2):
            for k in (8, 3):
                for tpw in TPW:
                    ctx.set_option("kernel", k)
                    ctx.set_option("tiles_per_wave", tpw)
                    t0 = time.perf_counter()
                    while time.perf_counter() - t0 < 0.4:
                        for _ in range(50):
                            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                        ctx.synchronize()
                    ts = []
                    for _ in range(200):
                        ctx.timer_mark(0)
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                        ctx.timer_mark(1)
                        ts.append(ctx.timer_between_ms(0, 1))
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(300):
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                    ctx.synchronize()
                    b2b = (time.perf_counter() - t0) / 300 * 1e3
                    got = np.empty(W * H, np.uint8)
                    ctx.d2h(got, d_m)
                    bad = int(np.count_nonzero(got != expect.reshape(-1)))
                    print(f"{cfg} {ctx.last_kernel_name()} tiles_per_wave {tpw}: median {np.median(ts):.4f} ms, min {np.min(ts):.4f}, back to back {b2b:.4f}; "
                          f"{bad} bytes differ from the oracle", flush=True)
