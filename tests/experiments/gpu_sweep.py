"""Experiment driver (GPU box): every kernel variant x option on the given configs, each checked
bit-for-bit against the oracle, then timed with hipEvents.  Prints one line per combination.
Lives under tests/ because it uses the oracle as its checker (test infrastructure, never the product path)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cornell_256,atrium_1080p,city_4k")
    ap.add_argument("--variants", default="")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--options", default="", help="semicolon list of key=v1,v2 to sweep, e.g. xcd_swizzle=0,1")
    ap.add_argument("--directional", action="store_true")
    args = ap.parse_args()
    from raytracedshadows_amd import api, workloads
    import oracle

    sweeps = []
    for item in filter(None, args.options.split(";")):
        k, vs = item.split("=")
        sweeps.append((k, [int(v) for v in vs.split(",")]))

    for cfg in args.configs.split(","):
        if "@" in cfg:                                   # ad-hoc size: scene@WxH (point light, 1 sample)
            scene, wh = cfg.split("@")
            W, H = (int(v) for v in wh.split("x"))
            light, spp = "point", 1
        else:
            scene, W, H, light, spp = workloads.CONFIGS[cfg]
        wl = workloads.prepare(scene, W, H, light="directional" if args.directional else light, spp=spp, log=print,
                               radius=workloads.SOFT_RADIUS.get(cfg, 0.01), table=workloads.PER_PIXEL_TABLE.get(cfg, 0))
        t0 = time.time()
        want, V, L = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                        oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
        tcpu = time.time() - t0
        n = wl.rays
        bytes_alg = 32 * V + 16 * L + 17 * W * H
        print(f"[{cfg}] oracle: {tcpu:.2f}s ({n / tcpu / 1e6:.1f} Mrays/s, {oracle.max_threads()} threads) "
              f"V/ray {V / n:.2f} L/ray {L / n:.2f} alg bytes/ray {bytes_alg / n:.1f} lit {want.mean():.3f}", flush=True)
        with api.ShadowContext(0) as ctx:
            ctx.set_bvh(wl.packed)
            d_pos = ctx.malloc(wl.positions.nbytes)
            d_mask = ctx.malloc(W * H)
            ctx.h2d(d_pos, wl.positions)
            nvar = ctx.get_option("kernel_count")
            print(f"[{cfg}] wide copy: {ctx.get_option('wide_nodes')} nodes in {ctx.get_option('wide_levels')} levels, "
                  f"enclosed {ctx.get_option('bvh_enclosed')}", flush=True)
            variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(nvar))
            combos = [{}]
            for k, vs in sweeps:
                combos = [dict(c, **{k: v}) for c in combos for v in vs]
            for variant in variants:
                for combo in combos:
                    ctx.set_option("kernel", variant)
                    for k, v in combo.items():
                        ctx.set_option(k, v)
                    got = np.zeros((H, W), np.uint8)
                    ctx.h2d(d_mask, got)
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                    ctx.synchronize()
                    ctx.d2h(got, d_mask)
                    bad = int((got != want).sum())
                    for _ in range(args.warmup):
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                    times = []
                    for _ in range(args.frames):
                        ctx.timer_begin()
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                        ctx.timer_end()
                        times.append(ctx.timer_elapsed_ms())
                    med = float(np.median(times))
                    print(json.dumps({"config": cfg, "kernel": variant, "name": ctx.last_kernel_name(), "opts": combo,
                                      "mismatch": bad, "ms": round(med, 4), "ms_min": round(min(times), 4),
                                      "Grays_s": round(n / med / 1e6, 3),
                                      "alg_TBps": round(bytes_alg / med / 1e9, 3)}), flush=True)
            ctx.free(d_pos)
            ctx.free(d_mask)


if __name__ == "__main__":
    main()
