"""GPU box: can the long tiles be predicted from the G-buffer alone?  Per 8x8 tile: measured wave life (wave statistics of the plain
launch) against cheap features of its 64 texels -- spread of the positions (bounding-box diagonal), spread relative to the distance
from the camera, number of background texels.  Prints rank correlations and how much of the longest 1 % / 3 % of the tiles the top
k % of each feature would catch.   python tools/long_tile_predictor.py atrium_1080p 3"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracedshadows_amd import api, workloads
    cfg, kernel = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    bx, by = (W + 7) // 8, (H + 7) // 8
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        ctx.set_option("kernel", kernel)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        waves = bx * by
        ctx.set_option("wave_stats", waves)
        for _ in range(3):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
        ctx.free(d_pos); ctx.free(d_mask)
    life = np.zeros((by, bx))
    tx = (st[:, 3] >> np.uint64(48)).astype(np.int64); ty = ((st[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    life[ty, tx] = (rt[:, 1] - rt[:, 0]).astype(np.float64) / 100.0
    pos = np.zeros((by * 8, bx * 8, 3), np.float32)
    pos[:H, :W] = wl.positions.reshape(H, W, 4)[..., :3]
    t = pos.reshape(by, 8, bx, 8, 3).transpose(0, 2, 1, 3, 4).reshape(by, bx, 64, 3)
    lo, hi = t.min(2), t.max(2)
    diag = np.linalg.norm(hi - lo, axis=2)
    dist = np.linalg.norm(t, axis=3)                      # camera-relative positions: distance from the eye
    rel = diag / np.maximum(dist.mean(2), 1e-6)
    depth_spread = dist.max(2) / np.maximum(dist.min(2), 1e-6)
    background = (np.abs(t).sum(3) == 0).sum(2)
    feats = {"bbox diagonal": diag, "diagonal / mean distance": rel, "max / min distance": depth_spread, "background texels": background.astype(np.float64)}
    L = life.ravel()
    order = np.argsort(-L)
    print(f"[{cfg}] kernel {kernel}: {L.size} tiles, life mean {L.mean():.1f} p99 {np.percentile(L, 99):.1f} max {L.max():.1f} us")
    for name, f in feats.items():
        F = f.ravel()
        rl, rf = np.argsort(np.argsort(L)), np.argsort(np.argsort(F))
        rho = np.corrcoef(rl, rf)[0, 1]
        line = f"   {name:26s} rank correlation with life {rho:+.2f};"
        fo = np.argsort(-F)
        for top in (0.01, 0.03):
            n = int(L.size * top)
            want = set(order[:n].tolist())
            for k in (0.03, 0.10, 0.33):
                got = set(fo[:int(L.size * k)].tolist())
                line += f" longest {top * 100:.0f} % within top {k * 100:.0f} % of the feature: {len(want & got) / n * 100:.0f} %;"
        print(line)


if __name__ == "__main__":
    main()
