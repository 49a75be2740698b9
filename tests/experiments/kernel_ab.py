"""Experiment (GPU box): kernels side by side on BASELINE configs in one process -- per kernel the median and minimum of 200
single launches between HIP events, 300 launches back to back, and the number of mask bytes that differ from the oracle's.
    KERNELS=3,8 python tests/experiments/kernel_ab.py city_4k courtyard_4k atrium_1080p city_4k_soft16"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracedshadows_amd import api, workloads
import oracle as orc

KERNELS = [int(v) for v in os.environ.get("KERNELS", "3,8").split(",")]
OPTS = [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]      # context options, e.g. OPTS=defer_pairs=32
for cfg in sys.argv[1:] or ["city_4k"]:
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    expect = orc.shadow_mask(wl.packed, wl.constants.as_array(), orc.light_from_product(wl.light, wl.constants), wl.positions, W, H)[0]
    n = int(os.environ.get("N", 40 if wl.spp > 1 else 200))
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        for k_, v_ in OPTS:
            ctx.set_option(k_, int(v_))
        d_pos, d_m = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for rep in range(2):
            for k in KERNELS:
                ctx.set_option("kernel", k)
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.4:
                    for _ in range(10):
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                    ctx.synchronize()
                ts = []
                for _ in range(n):
                    ctx.timer_mark(0)
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                    ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                ctx.synchronize()
                b2b = (time.perf_counter() - t0) / n * 1e3
                got = np.empty(W * H, np.uint8)
                ctx.d2h(got, d_m)
                bad = int(np.count_nonzero(got != expect.reshape(-1)))
                print(f"{cfg} kernel {k} {ctx.last_kernel_name()}: median {np.median(ts):.4f} ms, min {np.min(ts):.4f}, back to back {b2b:.4f} "
                      f"= {wl.rays / b2b / 1e6:.1f} Grays/s; {bad} bytes differ from the oracle", flush=True)
