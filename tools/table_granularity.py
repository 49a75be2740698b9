"""GPU box: what of a whole-dispatch table (front_share 1) survives a camera that moves, by the GRANULARITY of the measured lives it
is sorted by.  For a config: wave statistics of camera A's frame; tables planned from (a) those lives per 8x8 tile, (b) the
largest life of the B x B-tile block a tile lies in (B = 2 ... 32: the order only knows blocks), (c) one life for every tile (no
sort at all: what remains is the XCD placement of xcd_square, a static property of the frame size); each with and without
xcd_square.  Every table is timed on camera A's frame and, UNCHANGED, on cameras advanced 0.1 / 1 / 5 % along the view direction,
against the plain launch measured right beside it (same clocks).  Masks are checked against the CPU oracle on the last camera.
    python tools/table_granularity.py courtyard_4k city_4k"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from raytracedshadows_amd import api, workloads
    import oracle
    kernel = int(os.environ.get("KERNEL", 8))
    for cfg in sys.argv[1:] or ["courtyard_4k"]:
        wl = workloads.prepare_config(cfg, cache=True)
        W, H, sc = wl.W, wl.H, wl.scene
        bx_n, by_n = (W + 7) // 8, (H + 7) // 8
        waves = bx_n * by_n
        with api.ShadowContext(0) as ctx:
            ctx.set_bvh(wl.packed)
            ctx.set_option("kernel", kernel)
            ctx.set_option("packet_share", 4)
            d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)

            def frame(advance):
                eye = (sc.eye + (sc.target - sc.eye) * np.float32(advance)).astype(np.float32)
                pos = wl.positions if advance == 0 else api.primary_positions(wl.packed, eye, sc.target, sc.fovy, W, H)[0]
                return api.RayTracingConstants.make(eye, sc.light_direction, W, H), pos

            def go(k):
                ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=wl.light)

            def median(k, n=60):
                ts = []
                for _ in range(n):
                    ctx.timer_mark(0); go(k); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                return float(np.median(ts))

            def paired(k):
                """table / plain, both measured in this order after a warm-up with the table"""
                for _ in range(200):
                    go(k)
                t = median(k)
                ctx.set_option("tile_splits", 0)
                p = median(k)
                ctx.set_option("tile_splits", 1)
                return t, p

            cams = [(name, *frame(a)) for name, a in (("A", 0.0), ("+0.1 %", 0.001), ("+1 %", 0.01), ("+5 %", 0.05))]
            kA, posA = cams[0][1], cams[0][2]
            ctx.h2d(d_pos, posA)
            for _ in range(50):
                go(kA)
            ctx.set_option("wave_stats", waves)
            go(kA); go(kA); ctx.synchronize()
            st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
            ctx.set_option("wave_stats", 0)
            life = (rt[:, 1] - rt[:, 0]).astype(np.int64)                       # 10 ns units
            tx, ty = (st[:, 3] >> np.uint64(48)).astype(np.int64), ((st[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
            grid = np.zeros((by_n, bx_n), np.int64)
            grid[ty, tx] = life
            print(f"[{cfg}] kernel {kernel}: {waves} tiles, lives {np.percentile(life, 50) / 100:.1f} us median, {np.percentile(life, 99) / 100:.1f} p99, "
                  f"{life.max() / 100:.1f} longest", flush=True)

            def stats_for(lives_grid):
                r = rt.copy()
                r[:, 0] = 1000
                r[:, 1] = (1000 + np.maximum(1, lives_grid[ty, tx])).astype(np.uint64)
                return st, r

            variants = [("per tile", grid)]
            for B in (2, 4, 8, 16, 32):
                g = np.zeros_like(grid)
                for y0 in range(0, by_n, B):
                    for x0 in range(0, bx_n, B):
                        g[y0:y0 + B, x0:x0 + B] = grid[y0:y0 + B, x0:x0 + B].max()
                variants.append((f"block {B}x{B} max", g))
            variants.append(("no sort", np.full_like(grid, 1000)))
            for name, g in variants:
                for square in (0, 32):
                    ctx.h2d(d_pos, posA)
                    tiles, _ = ctx.plan_splits(kA, d_pos, W, H, d_mask, light=wl.light, min_life_us=1e9, piece_us=1e9, front_share=1.0,
                                               xcd_square=square, prev=stats_for(g))
                    assert tiles == waves, (tiles, waves)
                    row = []
                    for cname, k, pos in cams:
                        ctx.h2d(d_pos, pos)
                        t, p = paired(k)
                        row.append(f"{cname}: {t:.4f} / {p:.4f} ms ({(t / p - 1) * 100:+.1f} %)")
                    print(f"[{cfg}] sorted by life {name}, xcd_square {square}: " + "; ".join(row), flush=True)
            # the last table on the last camera: the mask
            cname, k, pos = cams[-1]
            want, _, _ = oracle.shadow_mask(wl.packed, k.as_array(), oracle.light_from_product(wl.light, k), pos, W, H)
            ctx.h2d(d_mask, np.full(W * H, 9, np.uint8))
            go(k); ctx.synchronize()
            got = np.empty((H, W), np.uint8)
            ctx.d2h(got, d_mask)
            print(f"[{cfg}] camera {cname} through the last table: {int((got != want).sum())} mask bytes differ from the oracle", flush=True)
            ctx.free(d_pos); ctx.free(d_mask)


if __name__ == "__main__":
    main()
