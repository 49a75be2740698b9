"""Diagnostics (GPU box): the longest waves of a frame -- tile, duration, whether the packet dissolved -- for a few
dissolve thresholds.  Writes gpurun_out/long_waves_<config>.json (joined with oracle statistics offline)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="atrium_1080p")
    ap.add_argument("--shares", default="4")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--kernel", type=int, default=3)
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config)
    W, H = wl.W, wl.H
    out = {}
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos = ctx.malloc(wl.positions.nbytes)
        d_mask = ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", args.kernel)
        waves = ((W + 7) // 8) * ((H + 7) // 8)
        for share in [int(s) for s in args.shares.split(",")]:
            ctx.set_option("packet_share", share)
            ctx.set_option("wave_stats", 0)
            ms = []
            for i in range(30):
                ctx.timer_begin()
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ctx.timer_end()
                if i >= 5:
                    ms.append(ctx.timer_elapsed_ms())
            ctx.set_option("wave_stats", waves)
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            st = ctx.read_wave_stats(waves)
            ctx.set_option("wave_stats", 0)
            t0, t1 = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
            left = -(st[:, 2] & np.uint64(1)).astype(np.int64)
            iters = ((st[:, 2] >> np.uint64(8)) & np.uint64(0xffffff)).astype(np.int64)
            tdis = (st[:, 2] >> np.uint64(32)).astype(np.float64) / 2400.0
            lsteps = (st[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
            left = np.where(left > 2**31, left - 2**32, left)
            dur = (t1 - t0) / 2400.0            # s_memtime counts shader cycles (~2.4 GHz): rough us
            end = (t1 - t0.min()) / 2400.0
            order = np.argsort(dur)[::-1][:args.top]
            rows = [{"bx": int(st[i, 3] >> np.uint64(48)), "by": int((st[i, 3] >> np.uint64(32)) & np.uint64(0xffff)),
                     "dur_us": round(float(dur[i]), 1), "end_us": round(float(end[i]), 1), "dissolved": bool(left[i] < 0),
                     "iters": int(iters[i]), "lane_steps": int(lsteps[i]), "packet_us": round(float(tdis[i]), 1)}
                    for i in order]
            dz = left < 0
            if dz.any():
                print(f"    dissolved waves: iterations mean {iters[dz].mean():.0f} max {iters[dz].max()}; lane occupancy of the "
                      f"lane-per-ray part {lsteps[dz].sum() / (64.0 * iters[dz].sum()):.2f}; us per iteration (top 40 waves) "
                      f"{np.mean([dur[i] / max(1, iters[i]) for i in order if dz[i]]):.3f}")
            out[str(share)] = {"ms_median": float(np.median(ms)), "ms_min": float(np.min(ms)),
                               "dissolved_waves": int((left < 0).sum()), "waves": int(waves),
                               "span_us": float(end.max()), "top": rows}
            print(f"share {share}: {np.median(ms):.4f} ms (min {np.min(ms):.4f}); dissolved {int((left < 0).sum())}/{waves}; "
                  f"span {end.max():.1f} us; top: " +
                  ", ".join(f"({r['bx']},{r['by']}) {r['dur_us']}us{'D' if r['dissolved'] else 'P'} it{r['iters']} ls{r['lane_steps']} pk{r['packet_us']}us" for r in rows[:8]), flush=True)
        ctx.free(d_pos)
        ctx.free(d_mask)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"long_waves_{args.config}.json"), "w"))


if __name__ == "__main__":
    main()
