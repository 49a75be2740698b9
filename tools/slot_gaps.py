"""Diagnostics (GPU box): how long does a wave slot stay EMPTY between two waves, and how evenly are the CUs filled?
Per-wave stamps (100 MHz clock at start / end, XCC id, HW_ID = wave slot / SIMD / CU / SE) of one frame of a packet kernel;
waves are grouped by hardware slot (XCC, HW_ID without the per-dispatch fields) and sorted by start time."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--kernel", type=int, default=8)
    ap.add_argument("--options", default="")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    waves = ((W + 7) // 8) * ((H + 7) // 8)
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", args.kernel)
        for kv in filter(None, args.options.split(",")):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))

        def go():
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        for _ in range(300):
            go()
        ctx.synchronize()
        ctx.set_option("wave_stats", waves)
        for _ in range(8):
            go()
        ctx.synchronize()
        rt = ctx.read_wave_realtime(waves)
        ctx.set_option("wave_stats", 0)
        ctx.free(d_pos)
        ctx.free(d_mask)
    ok = rt[:, 1] > rt[:, 0]
    t0 = (rt[ok, 0] - rt[ok, 0].min()).astype(np.float64) / 100.0          # us
    t1 = (rt[ok, 1] - rt[ok, 0].min()).astype(np.float64) / 100.0
    xcc = (rt[ok, 3] & np.uint64(0xF)).astype(np.int64)
    hw = (rt[ok, 3] >> np.uint64(32)).astype(np.int64)
    span = t1.max()
    fields = {"wave_id [3:0]": hw & 0xF, "simd_id [5:4]": (hw >> 4) & 3, "pipe [7:6]": (hw >> 6) & 3, "cu_id [11:8]": (hw >> 8) & 0xF,
              "sh_id [12]": (hw >> 12) & 1, "se_id [15:13]": (hw >> 13) & 7, "tg_id [19:16]": (hw >> 16) & 0xF}
    print(f"[{args.config}] kernel {args.kernel} {args.options}: {ok.sum()} waves, span {span:.1f} us, mean life {np.mean(t1 - t0):.2f} us")
    print("    distinct values per HW_ID field: " + ", ".join(f"{k}: {len(np.unique(v))}" for k, v in fields.items()))
    slot = (xcc << 16) | (hw & 0xFFFF)                                       # XCC + SE / SH / CU / SIMD / wave slot
    cu = (xcc << 16) | (hw & 0xFF00)
    simd = (xcc << 16) | (hw & 0xFF30)
    print(f"    distinct slots {len(np.unique(slot))}, SIMDs {len(np.unique(simd))}, CUs {len(np.unique(cu))}")
    order = np.lexsort((t0, slot))
    s_sorted, a, b = slot[order], t0[order], t1[order]
    same = s_sorted[1:] == s_sorted[:-1]
    gaps = (a[1:] - b[:-1])[same]
    steady = same & (a[1:] > 0.1 * span) & (a[1:] < 0.85 * span)
    g2 = (a[1:] - b[:-1])[steady]
    print(f"    gap between two waves of one slot: mean {gaps.mean():.2f} us, p10 {np.percentile(gaps, 10):.2f}, p50 {np.percentile(gaps, 50):.2f}, "
          f"p90 {np.percentile(gaps, 90):.2f}, p99 {np.percentile(gaps, 99):.2f}; in the steady part of the frame: mean {g2.mean():.2f}, p50 {np.percentile(g2, 50):.2f}")
    per_slot = np.bincount(np.unique(slot, return_inverse=True)[1])
    print(f"    waves per slot: mean {per_slot.mean():.1f}, min {per_slot.min()}, max {per_slot.max()}")
    mid = 0.5 * span
    infl = (t0 <= mid) & (t1 > mid)
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1][infl], minlength=len(np.unique(cu)))
    print(f"    waves in flight per CU at mid-frame: mean {per_cu.mean():.1f} of 32, min {per_cu.min()}, p10 {np.percentile(per_cu, 10):.0f}, "
          f"p50 {np.percentile(per_cu, 50):.0f}, p90 {np.percentile(per_cu, 90):.0f}, max {per_cu.max()}")
    per_simd = np.bincount(np.unique(simd, return_inverse=True)[1][infl], minlength=len(np.unique(simd)))
    print(f"    waves in flight per SIMD at mid-frame: mean {per_simd.mean():.2f} of 8, histogram 0..8: {np.bincount(per_simd, minlength=9).tolist()}")
    # busy time of a slot / span in the steady part
    lo, hi = 0.1 * span, 0.85 * span
    busy = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / ((hi - lo) * len(np.unique(slot)))
    print(f"    slot occupancy in the steady part of the frame: {busy:.2f}")


if __name__ == "__main__":
    main()
