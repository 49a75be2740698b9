"""GPU box (one GPU): what the 1/2/4/8-GPU striped frame (BASELINE configs[3]) will cost per device.  Every stripe of an
N-way partition is traced ALONE on this device, exactly as rank r of `bench.py --gpus N` would trace it (interleaved 32-row
bands, one dispatch); the slowest stripe bounds the frame.  Prints the predicted strong-scaling efficiency
t(1) / (N * max_r t(N, r)) -- a prediction from single-device timings, not a multi-GPU measurement."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--band", type=int, default=32)
    ap.add_argument("--kernel", type=int, default=-2, help="kernel id; -1 = library default; -2 = rts_ctx_autotune on the full frame (as bench.py does)")
    ap.add_argument("--options", default="", help="comma list of key=value context options")
    ap.add_argument("--tune-stripes", action="store_true",
                    help="every stripe tuned on its own dispatch (rts_ctx_autotune_stripes: kernel, dissolve threshold, split table) -- "
                         "what rank r of `bench.py --gpus N` does since round 4")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for kv in filter(None, args.options.split(",")):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        if args.kernel == -2:
            for _ in range(100):
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            print(f"[{args.config}] autotune: kernel {ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light)}")
        else:
            ctx.set_option("kernel", args.kernel)
        base = None
        for n in (1, 2, 4, 8):
            worst, times = 0.0, []
            tuned = []
            for r in range(n):
                def go():
                    ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, args.band, n, r, light=wl.light)
                for _ in range(300):
                    go()
                if args.tune_stripes:
                    ctx.set_option("packet_share", 4)
                    kid, _ = ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light, stripes=(args.band, n, r))
                    tuned.append(f"k{kid}/s{ctx.get_option('packet_share')}/{ctx.get_option('split_tiles')}t"
                                 + ("/ordered" if ctx.get_option("tile_order_tiles") else ""))
                    for _ in range(100):
                        go()
                ts = []
                for _ in range(40):
                    ctx.timer_mark(0); go(); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                times.append(float(np.median(ts)))
            worst = max(times)
            base = base or worst
            print(f"[{args.config}] band {args.band}, {n} stripe(s): per-stripe ms {' '.join(f'{t:.4f}' for t in times)}; slowest {worst:.4f} ms "
                  f"-> {wl.rays / worst / 1e6:.1f} Grays/s aggregate, predicted efficiency {base / (n * worst) * 100:.0f} % ({ctx.last_kernel_name()})"
                  + (f" tuned per stripe (kernel/share/split tiles[/ordered = a planned tile order]): {' '.join(tuned)}" if tuned else ""), flush=True)
        ctx.free(d_pos)
        ctx.free(d_mask)


if __name__ == "__main__":
    main()
