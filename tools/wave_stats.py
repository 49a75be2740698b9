"""Diagnostics (GPU box): per-wave start/end clocks of the packet kernels -> where the kernel time goes."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--kernels", default="3,4,5")
    ap.add_argument("--budgets", default="96")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config)
    W, H = wl.W, wl.H
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos = ctx.malloc(wl.positions.nbytes)
        d_mask = ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for kern in [int(k) for k in args.kernels.split(",")]:
            for budget in [int(b) for b in args.budgets.split(",")]:
                ctx.set_option("kernel", kern)
                ctx.set_option("packet_budget", budget)
                bwv = ctx.get_option("block_waves")
                f = 1 if bwv == 1 else 2
                tw = {3: 8, 4: 16, 5: 16}[kern] * f
                th = {3: 8, 4: 8, 5: 16}[kern] * f
                waves = ((W + tw - 1) // tw) * ((H + th - 1) // th) * bwv
                ctx.set_option("wave_stats", 0)
                for _ in range(3):
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ctx.synchronize()
                ctx.timer_begin()
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ctx.timer_end()
                ms_plain = ctx.timer_elapsed_ms()
                ctx.set_option("wave_stats", waves)
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ctx.synchronize()
                st = ctx.read_wave_stats(waves)
                ctx.set_option("wave_stats", 0)
                t0, t1 = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
                ok = t1 > 0
                t0, t1, left = t0[ok], t1[ok], -(st[ok, 2] & np.uint64(1)).astype(np.int64)
                left = np.where(left > 2**31, left - 2**32, left)
                base = t0.min()
                dur = t1 - t0
                span = t1.max() - base
                # s_memtime ticks at 100 MHz on gfx9 (constant clock) -> 10 ns per tick
                tick_us = 1.0 / 2400.0   # s_memtime counts shader cycles (~2.4 GHz): rough conversion
                order = np.argsort(dur)[::-1]
                used = budget - left
                print(f"[{args.config}] kernel {kern} budget {budget}: {ms_plain:.4f} ms plain; {ok.sum()} waves; "
                      f"span {span * tick_us:.1f} us; wave dur mean {dur.mean() * tick_us:.2f} us, p50 {np.percentile(dur, 50) * tick_us:.2f}, "
                      f"p99 {np.percentile(dur, 99) * tick_us:.2f}, max {dur.max() * tick_us:.2f} us; "
                      f"sum/8192slots {dur.sum() * tick_us / 8192:.1f} us; solo waves {(left < 0).sum()}; "
                      f"side-steps used mean {used.mean():.1f} max {used.max()}")
                late = (t1 - base) * tick_us
                for q in (50, 90, 99, 99.9, 100):
                    print(f"    {q}% of waves finished by {np.percentile(late, q):.1f} us")
                top = order[:5]
                print("    longest waves: " + ", ".join(
                    f"dur {dur[i] * tick_us:.1f}us start {(t0[i] - base) * tick_us:.1f}us used {used[i]} blk({int(st[ok][i, 3] >> np.uint64(48))},{int((st[ok][i, 3] >> np.uint64(32)) & np.uint64(0xffff))})"
                    for i in top))
        ctx.free(d_pos)
        ctx.free(d_mask)


if __name__ == "__main__":
    main()
