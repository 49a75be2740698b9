"""GPU box: how long the pieces of a split table live, against the tiles' own waves (rts_ctx_plan_splits).
    python tools/piece_stats.py --config atrium_1080p --kernel 3 --life 30 --end 0.5 --piece 15 --max-pieces 8"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="atrium_1080p")
    ap.add_argument("--kernel", type=int, default=3)
    ap.add_argument("--life", type=float, default=30.0)
    ap.add_argument("--end", type=float, default=0.5, help="fraction of the plain frame time")
    ap.add_argument("--piece", type=float, default=15.0)
    ap.add_argument("--max-pieces", type=int, default=8)
    ap.add_argument("--options", default="")
    ap.add_argument("--stripe", default="", help="n:r = the dispatch of stripe r of an n-way interleaved partition (32-row bands) instead of the frame")
    ap.add_argument("--tiles", type=int, default=0, help="max_tiles of the plan (0 = the library's 4096)")
    ap.add_argument("--share", type=float, default=0.0, help="front_share: the longest share of all tiles starts first (1 = the whole dispatch in table order)")
    ap.add_argument("--xcd", type=int, default=0, help="xcd_square")
    ap.add_argument("--front", type=float, default=0.0, help="front_life_us: tiles that lived longer start first, unsplit")
    args = ap.parse_args()
    stripes = (32, int(args.stripe.split(":")[0]), int(args.stripe.split(":")[1])) if args.stripe else None
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        ctx.set_option("kernel", args.kernel)
        for kv in filter(None, args.options.split(",")):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)

        def go():
            if stripes:
                ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, stripes[0], stripes[1], stripes[2], light=wl.light)
            else:
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)

        def med(n=60):
            for _ in range(200):
                go()
            ts = []
            for _ in range(n):
                ctx.timer_mark(0); go(); ctx.timer_mark(1)
                ts.append(ctx.timer_between_ms(0, 1))
            return float(np.median(ts))

        plain = med()
        waves = ((W + 7) // 8) * ((H + 7) // 8)
        if stripes:
            bands = (H + 31) // 32
            waves = ((W + 7) // 8) * (((bands - stripes[2] + stripes[1] - 1) // stripes[1]) * 4)
        ctx.set_option("wave_stats", waves)
        go(); go(); ctx.synchronize()
        st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
        ctx.set_option("wave_stats", 0)
        t0 = rt[:, 0].min()
        life = (rt[:, 1] - rt[:, 0]).astype(np.float64) / 100.0
        start = (rt[:, 0] - t0).astype(np.float64) / 100.0
        end = (rt[:, 1] - t0).astype(np.float64) / 100.0
        busy = life.sum() / 8192.0
        print(f"[{args.config}{' stripe ' + args.stripe if stripes else ''}] {waves} waves, sum of wave lives / 8192 slots = {busy:.1f} us")
        print(f"[{args.config}] kernel {args.kernel}: plain frame {plain * 1e3:.1f} us; with wave statistics: last wave starts at {start.max():.1f} us, "
              f"last ends at {end.max():.1f} us; wave life mean {life.mean():.1f} p50 {np.percentile(life, 50):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
        tiles, pieces = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=args.life, piece_us=args.piece,
                                        max_pieces=args.max_pieces, end_after_us=args.end * plain * 1e3, stripes=stripes, max_tiles=args.tiles, front_life_us=args.front,
                                        front_share=args.share, xcd_square=args.xcd)
        if not pieces:
            print("no tiles selected")
            return
        ctx.set_option("piece_stats", pieces)
        split = med()
        go(); ctx.synchronize()
        rec, clk = ctx.read_piece_stats(pieces)
        is_piece = (rec[:, 3] >> 24) > 0                       # (front tiles are walked by the everyday path: no stamps)
        print(f"records: {int(is_piece.sum())} pieces of {ctx.get_option('split_tiles')} split tiles, {ctx.get_option('front_tiles')} front tiles")
        rec, clk = rec[is_piece], clk[is_piece]
        if not len(rec):
            ctx.free(d_pos); ctx.free(d_mask)
            return
        p0 = clk[:, 0].min()
        plife = (clk[:, 1] - clk[:, 0]).astype(np.float64) / 100.0
        pend = (clk[:, 1] - p0).astype(np.float64) / 100.0
        pstart = (clk[:, 0] - p0).astype(np.float64) / 100.0
        print(f"table: {tiles} tiles, {pieces} pieces; frame {split * 1e3:.1f} us ({(split / plain - 1) * 100:+.1f} %)")
        print(f"pieces: start p50 {np.percentile(pstart, 50):.1f} max {pstart.max():.1f} us; life mean {plife.mean():.1f} p50 {np.percentile(plife, 50):.1f} "
              f"p90 {np.percentile(plife, 90):.1f} max {plife.max():.1f} us; last piece ends at {pend.max():.1f} us")
        f = lambda a: a.astype(np.float64) / 100.0
        setup, packet = f(clk[:, 2] - clk[:, 0]), f(np.maximum(clk[:, 3], clk[:, 2]) - clk[:, 2])
        diss = clk[:, 4] > 0
        handover = f(clk[diss, 4] - clk[diss, 3])
        lane = f(clk[diss, 5] - clk[diss, 4])
        tail = f(clk[:, 1] - clk[:, 5])
        nodes, entries = (clk[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.float64), (clk[:, 6] >> np.uint64(32)).astype(np.float64)
        print(f"where a piece's time goes (means, us): ray set-up {setup.mean():.2f}; packet phase {packet.mean():.2f} over {nodes.mean() / 4.0:.1f} calls of the loop (<= 4 pops each); {diss.mean() * 100:.0f} % dissolve: hand-over {handover.mean() if diss.any() else 0:.2f} "
              f"({entries[diss].mean() if diss.any() else 0:.1f} stack entries), lane phase {lane.mean() if diss.any() else 0:.2f}; closing atomics {tail.mean():.2f}")
        # per tile: the tile's own wave against its pieces
        key = rec[:, 0]
        tile_life = {}
        bx = (st[:, 3] >> np.uint64(48)).astype(np.int64)
        by = ((st[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
        for i in np.argsort(-life)[:4096]:
            tile_life[int(bx[i]) | (int(by[i]) << 16)] = life[i]
        rows = []
        for t in np.unique(key):
            m = key == t
            rows.append((tile_life.get(int(t), float("nan")), int(m.sum()), plife[m].max(), plife[m].sum(), pend[m].max()))
        rows.sort(reverse=True)
        print("longest tiles: own wave us -> pieces, longest piece us, sum of pieces us, last piece ends at")
        for r in rows[:12]:
            print(f"   {r[0]:7.1f} -> {r[1]:2d} pieces, longest {r[2]:6.1f}, sum {r[3]:7.1f}, end {r[4]:6.1f}")
        own = np.array([r[0] for r in rows]); longest = np.array([r[2] for r in rows]); total = np.array([r[3] for r in rows])
        ok = np.isfinite(own)
        print(f"over {ok.sum()} tiles: own wave {own[ok].sum():.0f} us in total, pieces {total[ok].sum():.0f} us in total ({total[ok].sum() / own[ok].sum():.2f} x); "
              f"longest piece / own wave: mean {np.mean(longest[ok] / own[ok]):.2f}")
        ctx.free(d_pos); ctx.free(d_mask)


if __name__ == "__main__":
    main()
