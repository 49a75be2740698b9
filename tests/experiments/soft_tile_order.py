"""Experiment (GPU box): 16-sample soft shadows (configs[4]: 4 waves per tile, no split table) with the tiles dispatched in the order
a whole-dispatch split table would use -- half-octave bands of measured life, longest first, each band dealt over the XCDs by image
squares (rtsh_split_front_order) -- through rts_ctx_set_tile_order.  Plain launch against ordered launch, same process; parity.
    python tests/experiments/soft_tile_order.py city_4k_soft16 courtyard_4k_soft16"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracedshadows_amd import api, workloads
import oracle as orc

for cfg in sys.argv[1:] or ["city_4k_soft16"]:
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    bxn, byn = (W + 7) // 8, (H + 7) // 8
    tiles = bxn * byn
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_m = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for kernel in (int(v) for v in os.environ.get("KERNELS", "3,8").split(",")):
            ctx.set_option("kernel", kernel)
            ctx.set_tile_order(None)

            def go():
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)

            def med(n=15):
                for _ in range(5):
                    go()
                ts = []
                for _ in range(n):
                    ctx.timer_mark(0); go(); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                return float(np.median(ts))

            plain = med()
            waves = tiles * 4                                      # (4 waves per tile workgroup: "soft_split")
            ctx.set_option("wave_stats", waves)
            go(); go(); ctx.synchronize()
            st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
            ctx.set_option("wave_stats", 0)
            ok = rt[:, 1] > rt[:, 0]
            life = np.where(ok, (rt[:, 1] - rt[:, 0]).astype(np.float64) / 100.0, 0.0)
            tx, ty = (st[:, 3] >> np.uint64(48)).astype(np.int64), ((st[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
            grid = np.zeros((byn, bxn))
            np.maximum.at(grid, (ty[ok], tx[ok]), life[ok])          # a tile is as long as its longest wave
            by, bx = np.divmod(np.arange(tiles), bxn)
            tile_id = (bx | (by << 16)).astype(np.uint32)
            for name, square, block in (("sorted per tile", 0, 0), ("sorted per tile, XCD squares 32", 32, 0), ("blocks of 16, XCD squares 32", 32, 16)):
                order = api.split_front_order(grid.reshape(-1).astype(np.float32), tile_id, xcd_square=square, life_block=block)
                ctx.set_tile_order(order.astype(np.uint32))
                t = med()
                ctx.h2d(d_m, np.full(W * H, 99, np.uint8))
                go(); ctx.synchronize()
                got = np.empty(W * H, np.uint8)
                ctx.d2h(got, d_m)
                ctx.set_tile_order(None)
                p2 = med()
                print(f"[{cfg}] kernel {kernel} ({ctx.last_kernel_name()}): {name}: {t:.4f} ms against {p2:.4f} plain beside it ({(t / p2 - 1) * 100:+.1f} %); first plain {plain:.4f}", flush=True)
                if "want" not in dir():
                    want = orc.shadow_mask(wl.packed, wl.constants.as_array(), orc.light_from_product(wl.light, wl.constants), wl.positions, W, H)[0]
                print(f"    {int(np.count_nonzero(got != want.reshape(-1)))} bytes differ from the oracle", flush=True)
        del want
        ctx.free(d_pos); ctx.free(d_m)
