"""Diagnostics (GPU box): cost of the dispatch itself -- same 4K frame, BVH of ONE far-away triangle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracedshadows_amd import api, workloads


def main():

    wl = workloads.prepare_config("city_4k", cache=True)
    W, H = wl.W, wl.H
    tri = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]], np.float32)
    one = api.BVHBuilder().build(tri, 3, np.arange(3, dtype=np.uint32), 1).m_packedNodes
    with api.ShadowContext(0) as ctx:
        d_pos = ctx.malloc(wl.positions.nbytes); d_mask = ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        for name, packed in (("one-triangle BVH", one), ("city BVH", wl.packed)):
            ctx.set_bvh(packed)
            for kern in (0, 3, 4, 5):
                for bw in (1, 4):
                    if kern == 0 and bw == 1:
                        continue
                    ctx.set_option("kernel", kern); ctx.set_option("block_waves", bw); ctx.set_option("packet_budget", 1000)
                    for _ in range(300):                  # clocks ramped before anything is timed
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
                    ts = []
                    for _ in range(20):
                        ctx.timer_begin(); ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light); ctx.timer_end()
                        ts.append(ctx.timer_elapsed_ms())
                    print(f"{name}: kernel {kern} block_waves {bw}: {np.median(ts):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
