"""Diagnostics (GPU box): where does a frame's time go before any traversal?  Per-wave stamps of the packet kernel
(start / first ray ready / end in shader clocks, start / end on the 100 MHz realtime counter, XCC id) for the same 4K frame
against (a) a BVH of ONE far-away triangle (the "dispatch floor"), (b) the real BVH.

Prints, per case: kernel time; shader clock; mean wave lifetime and its split (start -> ray ready -> end); average number
of waves in flight (sum of lifetimes / kernel time; 8192 = every slot of 1024 SIMDs x 8 busy); how fast waves are
launched (waves started per microsecond while the machine fills, and in steady state)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--block-waves", default="1")
    ap.add_argument("--kernel", type=int, default=3)
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    tri = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]], np.float32)
    one = api.BVHBuilder().build(tri, 3, np.arange(3, dtype=np.uint32), 1).m_packedNodes
    with api.ShadowContext(0) as ctx:
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", args.kernel)

        def go():
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)

        for bw in [int(x) for x in args.block_waves.split(",")]:
            ctx.set_option("block_waves", bw)
            f = 1 if bw == 1 else 2
            waves = ((W + 8 * f - 1) // (8 * f)) * ((H + 8 * f - 1) // (8 * f)) * bw
            for name, packed in (("one-triangle BVH", one), (f"{args.config} BVH", wl.packed)):
                ctx.set_bvh(packed)
                for _ in range(300):
                    go()
                ctx.synchronize()
                ts = []
                for _ in range(30):
                    ctx.timer_mark(0); go(); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                ms = float(np.median(ts))
                ctx.set_option("wave_stats", waves)
                for _ in range(8):
                    go()
                ctx.synchronize()
                st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
                ctx.set_option("wave_stats", 0)
                ok = st[:, 1] > st[:, 0]
                life = (st[ok, 1] - st[ok, 0]).astype(np.float64)
                ready = rt[ok, 2].astype(np.float64)
                r0, r1 = rt[ok, 0].astype(np.float64), rt[ok, 1].astype(np.float64)
                clock = life.sum() / (r1 - r0).sum() * 100.0                      # MHz
                span_us = (r1.max() - r0.min()) / 100.0
                in_flight = (r1 - r0).sum() / 100.0 / span_us
                start_us = np.sort(r0 - r0.min()) / 100.0
                k = min(8192, start_us.size) - 1
                fill_us = start_us[k]
                steady = (start_us.size - 1 - k) / max(1e-9, start_us[-1] - fill_us) if start_us.size > 8192 else float("nan")
                xcc = rt[ok, 3] & np.uint64(0xF)
                print(f"[{name}] block_waves {bw}: {ms * 1e3:.1f} us (diag span {span_us:.1f} us), clock {clock:.0f} MHz, "
                      f"{ok.sum()} waves: lifetime mean {life.mean() / clock:.2f} us (p50 {np.percentile(life, 50) / clock:.2f}, "
                      f"p99 {np.percentile(life, 99) / clock:.2f}, max {life.max() / clock:.2f}); start->ray ready mean "
                      f"{ready.mean() / clock:.2f} us ({ready.mean():.0f} clk, p50 {np.percentile(ready, 50):.0f}, p99 {np.percentile(ready, 99):.0f}); "
                      f"waves in flight avg {in_flight:.0f} of 8192; first 8192 waves started within {fill_us:.2f} us "
                      f"({(k + 1) / max(fill_us, 1e-9):.0f} waves/us), then {steady:.0f} waves/us; "
                      f"waves per XCC {np.bincount(xcc.astype(np.int64), minlength=8).tolist()}", flush=True)
                # timeline: waves in flight in 20 equal slices of the kernel, and where the waves that end last sit
                t0, t1 = (r0 - r0.min()) / 100.0, (r1 - r0.min()) / 100.0
                edges = np.linspace(0.0, span_us, 21)
                mid = (edges[:-1] + edges[1:]) / 2
                infl = [int(((t0 <= m) & (t1 > m)).sum()) for m in mid]
                print("    in flight per 5% slice: " + " ".join(str(v) for v in infl))
                xi = xcc.astype(np.int64)
                per_xcc = [float(((r1 - r0)[xi == x]).sum() / 100.0 / span_us) for x in range(8)]
                print("    mean waves in flight per XCC (1024 slots each): " + " ".join(f"{v:.0f}" for v in per_xcc))
                mid_t = 0.5 * span_us
                print("    waves in flight per XCC at mid-frame: " + " ".join(str(int(((t0 <= mid_t) & (t1 > mid_t) & (xi == x)).sum())) for x in range(8)))
                late = t1 > 0.85 * span_us
                ty = ((st[ok, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
                rows = ty[late]
                print(f"    waves ending in the last 15%: {late.sum()}, started at {np.percentile(t0[late], [0, 50, 100]).round(1).tolist()} us, "
                      f"lifetime {np.percentile((t1 - t0)[late], [0, 50, 100]).round(1).tolist()} us, tile rows {np.percentile(rows, [0, 50, 100]).tolist()} of {ty.max() + 1}")
                cost_by_band = [float((t1 - t0)[(ty >= a) & (ty < a + 27)].mean()) for a in range(0, int(ty.max()) + 1, 27)]
                print("    mean lifetime (us) per band of 27 tile rows, top to bottom: " + " ".join(f"{c:.1f}" for c in cost_by_band))
        ctx.free(d_pos)
        ctx.free(d_mask)


if __name__ == "__main__":
    main()
