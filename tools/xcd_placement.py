"""Experiment (GPU box): XCD-aware tile placement through the tile-order table.

Workgroups are dealt to the 8 XCDs round-robin (workgroup i -> XCD i % 8), each XCD has its own 4 MB L2.  With the
natural raster order every XCD sees every region of the image, i.e. the whole BVH.  Here XCD x gets the tiles of the
bands b with b % 8 == x (a band = R rows of tiles, or a square of S x S tiles dealt in a 2-D pattern), so that its L2
only has to hold the part of the tree under "its" image regions, while the bands are fine enough to balance the load.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracedshadows_amd import api, workloads


def main():


    def interleave(lists, n):
        """order[8*j + x] = lists[x][j]; leftovers (unequal lengths) fill the remaining slots round-robin."""
        order = np.full(n, -1, np.int64)
        rest = []
        for x, l in enumerate(lists):
            k = min(len(l), (n - x + 7) // 8)
            order[x::8][:k] = l[:k]
            rest.extend(l[k:])
        free = np.flatnonzero(order < 0)
        order[free] = rest
        assert sorted(order.tolist()) == list(range(n))
        return order.astype(np.uint32)


    def bands(bx, by, R):
        lists = [[] for _ in range(8)]
        for row in range(by):
            lists[(row // R) % 8].extend(range(row * bx, (row + 1) * bx))
        return interleave(lists, bx * by)


    def squares(bx, by, S):
        lists = [[] for _ in range(8)]
        for sy in range(0, by, S):
            for sx in range(0, bx, S):
                x = ((sx // S) + 3 * (sy // S)) % 8          # 2-D dealing: neighbours in x and y go to different XCDs
                for row in range(sy, min(by, sy + S)):
                    lists[x].extend(range(row * bx + sx, row * bx + min(bx, sx + S)))
        return interleave(lists, bx * by)


    for cfg in sys.argv[1:] or ("atrium_1080p", "city_4k"):
        wl = workloads.prepare_config(cfg)
        W, H = wl.W, wl.H
        with api.ShadowContext(0) as ctx:
            ctx.set_bvh(wl.packed)
            d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
            ctx.h2d(d_pos, wl.positions)
            ctx.set_option("kernel", 3)
            bx, by = (W + 7) // 8, (H + 7) // 8

            def timeit(tag):
                for _ in range(5):
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                ts = []
                for _ in range(60):
                    ctx.timer_begin(); ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light); ctx.timer_end()
                    ts.append(ctx.timer_elapsed_ms())
                got = np.zeros((H, W), np.uint8); ctx.d2h(got, d_mask)
                print(f"[{cfg}] {tag}: {np.median(ts):.4f} ms (min {min(ts):.4f})", flush=True)
                return got
            ref = timeit("natural order (no table)")
            ctx.set_tile_order(np.arange(bx * by, dtype=np.uint32))
            assert (timeit("natural order through the table") == ref).all()
            for R in (1, 2, 4, 8, 16):
                ctx.set_tile_order(bands(bx, by, R))
                assert (timeit(f"bands of {R} tile rows per XCD") == ref).all()
            for S in (4, 8, 16, 32):
                ctx.set_tile_order(squares(bx, by, S))
                assert (timeit(f"squares of {S}x{S} tiles per XCD") == ref).all()
            ctx.set_tile_order(None)
            ref2 = timeit("natural order again")
            ctx.free(d_pos); ctx.free(d_mask)


if __name__ == "__main__":
    main()
