"""Experiment (GPU box): split tables (rts_ctx_plan_splits) against the plain launch, same process, same clocks.
For every config and base kernel: the frame without a table, then with tables planned at several (min_life_us, piece_us,
max_pieces); per point the median / minimum of N launches between HIP events, back-to-back wall time, and the number of mask
bytes that differ from the oracle's.  A plan is life:piece:max_pieces[:end fraction[:front_life_us]];
STRIPE=band:n:r measures the dispatch of one interleaved stripe instead of the frame.
    KERNELS=3,8 PLANS=20:8:8,40:10:8:0.7 python tests/experiments/split_ab.py atrium_1080p courtyard_4k city_4k"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracedshadows_amd import api, workloads
import oracle as orc

KERNELS = [int(v) for v in os.environ.get("KERNELS", "3,8").split(",")]
PLANS = [tuple(float(x) for x in pl.split(":")) for pl in os.environ.get("PLANS", "30:10:8,20:6:8,15:5:16").split(",")]
OPTS = [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]
N = int(os.environ.get("N", 100))
STRIPE = tuple(int(v) for v in os.environ["STRIPE"].split(":")) if os.environ.get("STRIPE") else None     # band:n:r


def timeit(ctx, go, n):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            go()
        ctx.synchronize()
    ts = []
    for _ in range(n):
        ctx.timer_mark(0); go(); ctx.timer_mark(1)
        ts.append(ctx.timer_between_ms(0, 1))
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        go()
    ctx.synchronize()
    return float(np.median(ts)), float(np.min(ts)), (time.perf_counter() - t0) / n * 1e3


for cfg in sys.argv[1:] or ["atrium_1080p"]:
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    expect = orc.shadow_mask(wl.packed, wl.constants.as_array(), orc.light_from_product(wl.light, wl.constants), wl.positions, W, H)[0]
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        for k_, v_ in OPTS:
            ctx.set_option(k_, int(v_))
        d_pos, d_m = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)

        def go():
            if STRIPE:
                ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_m, STRIPE[0], STRIPE[1], STRIPE[2], light=wl.light)
            else:
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)

        def check():
            ctx.h2d(d_m, np.full(W * H, 9, np.uint8))
            go(); ctx.synchronize()
            got = np.empty(W * H, np.uint8)
            ctx.d2h(got, d_m)
            if STRIPE:
                from raytracedshadows_amd import partition
                own = np.zeros(H, bool)
                for b, e in partition.stripe_rows(H, STRIPE[1], STRIPE[2], band=STRIPE[0], interleaved=True):
                    own[b:e] = True
                return int(np.count_nonzero(got.reshape(H, W)[own] != expect[own])) + int(np.count_nonzero(got.reshape(H, W)[~own] != 9))
            return int(np.count_nonzero(got != expect.reshape(-1)))

        for k in KERNELS:
            ctx.set_option("kernel", k)
            ctx.clear_splits()
            med, mn, b2b = timeit(ctx, go, N)
            print(f"{cfg} kernel {k} no table: median {med:.4f} ms, min {mn:.4f}, back to back {b2b:.4f}; {check()} bytes differ ({ctx.last_kernel_name()})", flush=True)
            for life, piece, maxp, *rest in PLANS:
                endfrac = rest[0] if rest else 0.0           # tiles that ended later than this fraction of the plain frame
                front = rest[1] if len(rest) > 1 else 0.0     # front_life_us
                share = rest[2] if len(rest) > 2 else 0.0     # front_share
                t0 = time.perf_counter()
                tiles, pieces = ctx.plan_splits(wl.constants, d_pos, W, H, d_m, light=wl.light, min_life_us=life, piece_us=piece, max_pieces=int(maxp),
                                                end_after_us=endfrac * med * 1e3, front_life_us=front, front_share=share, stripes=STRIPE,
                                                max_tiles=int(os.environ.get("MAXTILES", 0)), xcd_square=int(os.environ.get("XCD_SQUARE", 0)))
                plan_ms = (time.perf_counter() - t0) * 1e3
                bad = check()
                med2, mn2, b2b2 = timeit(ctx, go, N)
                print(f"{cfg} kernel {k} table life>{life:g}us end>{endfrac:g}T piece {piece:g}us max {int(maxp)} front>{front:g}us/{share:g}: {tiles} tiles, {pieces} pieces (planned in {plan_ms:.0f} ms): "
                      f"median {med2:.4f} ms ({(med2 / med - 1) * 100:+.1f} %), min {mn2:.4f}, back to back {b2b2:.4f}; {bad} bytes differ", flush=True)
            ctx.clear_splits()
        ctx.free(d_pos); ctx.free(d_m)
