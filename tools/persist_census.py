"""Diagnostics (GPU box): the persistent grid (kernel 8) -- how many waves are resident at once, how many tiles each takes,
how the tiles spread over the XCCs."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="city_4k")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", 8)

        def go():
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)

        for _ in range(200):
            go()
        ctx.synchronize()
        ts = []
        for _ in range(30):
            ctx.timer_mark(0); go(); ctx.timer_mark(1)
            ts.append(ctx.timer_between_ms(0, 1))
        waves = 8192
        ctx.set_option("wave_stats", waves)
        for _ in range(5):
            go()
        ctx.synchronize()
        st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
        ctx.set_option("wave_stats", 0)
        ok = rt[:, 1] > rt[:, 0]
        r0, r1 = rt[ok, 0].astype(np.float64), rt[ok, 1].astype(np.float64)
        t0, t1 = (r0 - r0.min()) / 100.0, (r1 - r0.min()) / 100.0
        tiles = st[ok, 2].astype(np.int64)
        xcc = (st[ok, 3] >> np.uint64(32)).astype(np.int64)
        moves = (st[ok, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        clock = (st[ok, 1] - st[ok, 0]).astype(np.float64).sum() / (r1 - r0).sum() * 100.0
        span = t1.max()
        print(f"[{args.config}] kernel 8: {np.median(ts) * 1e3:.1f} us; {ok.sum()} waves recorded, span {span:.1f} us, clock {clock:.0f} MHz")
        print(f"    wave start times (us): p0 {t0.min():.1f} p50 {np.percentile(t0, 50):.1f} p90 {np.percentile(t0, 90):.1f} p99 {np.percentile(t0, 99):.1f} max {t0.max():.1f}")
        print(f"    wave end times (us):   p1 {np.percentile(t1, 1):.1f} p50 {np.percentile(t1, 50):.1f} p99 {np.percentile(t1, 99):.1f} max {t1.max():.1f}")
        print(f"    tiles per wave: min {tiles.min()} p50 {int(np.percentile(tiles, 50))} max {tiles.max()} sum {tiles.sum()}; moves per wave mean {moves.mean():.2f} max {moves.max()}")
        print(f"    waves per XCC {np.bincount(xcc, minlength=8).tolist()}; tiles per XCC {[int(tiles[xcc == k].sum()) for k in range(8)]}")
        edges = np.linspace(0.0, span, 21)
        mid = (edges[:-1] + edges[1:]) / 2
        print("    in flight per 5% slice: " + " ".join(str(int(((t0 <= m) & (t1 > m)).sum())) for m in mid))
        ctx.free(d_pos)
        ctx.free(d_mask)


if __name__ == "__main__":
    main()
