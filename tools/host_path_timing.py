"""GPU box: PCIe-inclusive rate of the host-pointer entry (rts_trace_shadow_mask: H2D positions, trace, D2H mask)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracedshadows_amd import api, workloads
wl = workloads.prepare_config("city_4k")
with api.ShadowContext(0) as ctx:
    ctx.set_bvh(wl.packed)
    out = np.zeros((wl.H, wl.W), np.uint8)
    for _ in range(3):
        ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light, out=out)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light, out=out)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print(f"host-pointer entry, city_4k: {t * 1e3:.2f} ms per frame = {wl.rays / t / 1e9:.2f} Grays/s "
          f"({wl.positions.nbytes / 1e6:.0f} MB in, {out.nbytes / 1e6:.1f} MB out, pageable host memory)")
