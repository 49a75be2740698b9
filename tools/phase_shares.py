"""GPU box: where the wave-time of a frame goes -- ray set-up, packet phase, lane-per-ray phase after a dissolve -- summed over all
waves (wave statistics of the diagnostic instantiation).   python tools/phase_shares.py courtyard_4k 8"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracedshadows_amd import api, workloads
    cfg, kernel = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        ctx.set_option("kernel", kernel)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        waves = ((W + 7) // 8) * ((H + 7) // 8)
        ctx.set_option("wave_stats", waves)
        for _ in range(3):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
        life_clk = (st[:, 1] - st[:, 0]).astype(np.float64)
        life_us = (rt[:, 1] - rt[:, 0]).astype(np.float64) / 100.0
        clk_per_us = life_clk.sum() / life_us.sum()
        ready = rt[:, 2].astype(np.float64) / clk_per_us
        dissolved = (st[:, 2] & np.uint64(1)) == 1
        t_diss = (st[:, 2] >> np.uint64(32)).astype(np.float64) / clk_per_us
        iters = ((st[:, 2] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.float64)
        lane = np.where(dissolved, np.maximum(life_us - t_diss, 0.0), 0.0)
        packet = life_us - ready - lane
        tot = life_us.sum()
        print(f"[{cfg}] kernel {kernel}: {waves} waves, sum of lives {tot / 1e3:.1f} ms ({tot / 8192:.1f} us per wave slot); shader clock {clk_per_us:.0f} MHz")
        print(f"   ray set-up (start -> first ray ready): {ready.sum() / tot * 100:.1f} %  ({ready.mean():.2f} us per wave)")
        print(f"   packet phase: {packet.sum() / tot * 100:.1f} %")
        print(f"   lane-per-ray phase: {lane.sum() / tot * 100:.1f} %  ({dissolved.sum()} waves dissolve = {dissolved.mean() * 100:.1f} %; "
              f"{iters[dissolved].mean() if dissolved.any() else 0:.0f} iterations each, {lane[dissolved].sum() / max(1.0, iters[dissolved].sum()):.2f} us per iteration)")
        ctx.free(d_pos); ctx.free(d_mask)


if __name__ == "__main__":
    main()
