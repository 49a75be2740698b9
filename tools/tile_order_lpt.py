"""Experiment (GPU box): does dispatching the longest tiles first shorten the kernel?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracedshadows_amd import api, workloads


def main():

    for cfg in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("atrium_1080p", "city_4k")):
        wl = workloads.prepare_config(cfg, cache=True)
        W, H = wl.W, wl.H
        with api.ShadowContext(0) as ctx:
            ctx.set_bvh(wl.packed)
            d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
            ctx.h2d(d_pos, wl.positions)
            ctx.set_option("kernel", 3)
            bx, by = (W + 7) // 8, (H + 7) // 8
            waves = bx * by

            def timeit(tag):
                for _ in range(400):                                   # (clocks up: a handful of launches is a cold measurement)
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ts = []
                for _ in range(40):
                    ctx.timer_mark(0); ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                got = np.zeros((H, W), np.uint8); ctx.d2h(got, d_mask)
                print(f"[{cfg}] {tag}: {np.median(ts):.4f} ms (min {min(ts):.4f})", flush=True)
                return got
            ref = timeit("natural order")
            ctx.set_option("wave_stats", waves)
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            st = ctx.read_wave_stats(waves)
            ctx.set_option("wave_stats", 0)
            dur = (st[:, 1] - st[:, 0]).astype(np.int64)          # block i == tile i in natural order
            order = np.argsort(-dur, kind="stable").astype(np.uint32)
            ctx.set_tile_order(order)
            a = timeit("longest tile first (measured durations)")
            assert (a == ref).all()
            rs = np.random.RandomState(0)
            ctx.set_tile_order(rs.permutation(waves).astype(np.uint32))
            b = timeit("random order")
            assert (b == ref).all()
            # coarse: 16 buckets by log2 duration, natural order inside a bucket
            bucket = np.clip(np.log2(np.maximum(dur, 1)).astype(np.int64), 0, 63)
            ctx.set_tile_order(np.argsort(-bucket, kind="stable").astype(np.uint32))
            c = timeit("longest first, log2 buckets")
            assert (c == ref).all()
            # rows of tiles in the order of their longest tile (natural order inside a row): what a per-row feedback could do
            rowmax = dur.reshape(by, bx).max(1)
            rows = np.argsort(-rowmax, kind="stable")
            ctx.set_tile_order((rows[:, None] * bx + np.arange(bx)[None, :]).reshape(-1).astype(np.uint32))
            d = timeit("rows by their longest tile, tiles in natural order")
            assert (d == ref).all()
            rowsum = dur.reshape(by, bx).sum(1)
            rows = np.argsort(-rowsum, kind="stable")
            ctx.set_tile_order((rows[:, None] * bx + np.arange(bx)[None, :]).reshape(-1).astype(np.uint32))
            e = timeit("rows by their total time, tiles in natural order")
            assert (e == ref).all()
            ctx.set_tile_order(np.arange(waves, dtype=np.uint32))
            f = timeit("natural order through the order table")
            assert (f == ref).all()
            ctx.set_tile_order(None)
            print(f"[{cfg}] durations: mean {dur.mean():.0f} p50 {np.percentile(dur,50):.0f} p99 {np.percentile(dur,99):.0f} max {dur.max()} cycles; sum/8192 = {dur.sum()/8192:.0f}")


if __name__ == "__main__":
    main()
