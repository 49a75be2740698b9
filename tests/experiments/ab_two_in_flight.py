import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from raytracedshadows_amd import api, workloads
cfg = sys.argv[1]
wl = workloads.prepare_config(cfg, cache=True)
W, H = wl.W, wl.H
with api.ShadowContext(0) as ctx:
    ctx.set_bvh(wl.packed)
    d_pos = ctx.malloc(wl.positions.nbytes); d_m = [ctx.malloc(W * H), ctx.malloc(W * H)]
    ctx.h2d(d_pos, wl.positions)
    st = [ctx.stream_create(), ctx.stream_create()]
    for k, opts in ((3, {}), (8, {"wide_lane": 0}), (8, {"wide_lane": 1}), (3, {}), (8, {"wide_lane": 0}), (8, {"wide_lane": 1})):
        ctx.set_option("kernel", k)
        for kk, v in opts.items(): ctx.set_option(kk, v)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            for _ in range(50): ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m[0], light=wl.light)
            ctx.synchronize()
        ts = []
        for _ in range(200):
            ctx.timer_mark(0); ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m[0], light=wl.light); ctx.timer_mark(1)
            ts.append(ctx.timer_between_ms(0, 1))
        # two in flight
        for i in range(20): ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m[i & 1], light=wl.light, stream=st[i & 1])
        ctx.synchronize(st[0]); ctx.synchronize(st[1])
        t0 = time.perf_counter()
        for i in range(400): ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m[i & 1], light=wl.light, stream=st[i & 1])
        ctx.synchronize(st[0]); ctx.synchronize(st[1])
        w2 = (time.perf_counter() - t0) / 400
        print(cfg, ctx.last_kernel_name(), opts, f"median {np.median(ts):.4f} ms, min {np.min(ts):.4f}; two in flight {w2*1e3:.4f} ms per frame", flush=True)
