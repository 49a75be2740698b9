"""Runs a few launches of one config with the given options (for rocprofv3 passes): python3 tools/run_launches.py city_4k kernel=3 block_waves=4"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracedshadows_amd import api, workloads
    cfg = sys.argv[1]
    opts = dict(kv.split("=") for kv in sys.argv[2:])
    n = int(opts.pop("launches", 12))
    split = opts.pop("split", None)                 # life_us:piece_us:max_pieces[:end fraction of a 0.15 ms frame]: plan a split table first
    wl = workloads.prepare_config(cfg, cache=True)
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        for k, v in opts.items():
            ctx.set_option(k, int(v))
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(wl.W * wl.H)
        ctx.h2d(d_pos, wl.positions)
        if split:
            f = [float(x) for x in split.split(":")]
            print("split table:", ctx.plan_splits(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light, min_life_us=f[0], piece_us=f[1],
                                                  max_pieces=int(f[2]), end_after_us=(f[3] if len(f) > 3 else 0.0)))
        for _ in range(n):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light)
        ctx.synchronize()
        print(ctx.last_kernel_name())
        ctx.free(d_pos)
        ctx.free(d_mask)


if __name__ == "__main__":
    main()
