#!/usr/bin/env python3
"""CPU count (with the oracle): how many rays take part in a box test of the LONGEST packets of a frame?
tools/long_packets.py [scene W H] ...   ("per-step member count of the never-dissolving courtyard packets")

Walks every 8x8 tile as the wide packet kernel does (orc_wide_packet_sim, cheap test + exact confirmation) and prints, for
all tiles and for the longest 1 % / 0.1 % / 8 of them, the wide nodes entered and the member lanes per wave-wide box test."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracedshadows_amd import workloads  # noqa: E402
import oracle  # noqa: E402

o = oracle._o
o.orc_wide_packet_sim.restype = None
o.orc_wide_packet_sim.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
o.orc_wide_packet_sim_per_tile.argtypes = [C.c_void_p] * 3


def main():
    args = sys.argv[1:] or ["courtyard", "3840", "2160", "city", "3840", "2160", "atrium", "1920", "1080"]
    for i in range(0, len(args), 3):
        scene, W, H = args[i], int(args[i + 1]), int(args[i + 2])
        wl = workloads.prepare(scene, W, H, via_obj=False)
        lt = oracle.light_from_product(wl.light, wl.constants)
        k = np.ascontiguousarray(wl.constants.as_array(), np.float32)
        packed = np.ascontiguousarray(wl.packed, np.uint32)
        pos = np.ascontiguousarray(wl.positions, np.float32)
        nt = (W // 8) * (H // 8)
        st, te, la = (np.zeros(nt, np.uint32) for _ in range(3))
        o.orc_wide_packet_sim_per_tile(oracle._p(st), oracle._p(te), oracle._p(la))
        out = np.zeros(16, np.uint64)
        o.orc_wide_packet_sim(oracle._p(packed), oracle._p(k), C.byref(lt), oracle._p(pos), W, H, 12, 16, oracle._p(out), None)
        o.orc_wide_packet_sim_per_tile(None, None, None)
        assert int(out[7]) == 0, "the wide walk differs from the reference walk"
        order = np.argsort(st)[::-1]
        for name, sel in (("all tiles", np.arange(nt)), ("longest 1 %", order[:nt // 100]),
                          ("longest 0.1 %", order[:max(1, nt // 1000)]), ("longest 8", order[:8])):
            m = la[sel].sum() / max(1, te[sel].sum())
            print(f"{scene} {W}x{H}: {name:14s}: wide nodes per tile {st[sel].mean():7.1f}, rays per box test {m:5.1f} of 64 (fill {m / 64:.2f})",
                  flush=True)


if __name__ == "__main__":
    main()
