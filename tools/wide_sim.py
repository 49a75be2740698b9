#!/usr/bin/env python3
"""CPU count (with the oracle) of what a packet walk over WIDE nodes would do: tools/wide_sim.py [scene W H] ...

Prints, per collapse depth (1 = both children's boxes in the parent, 2 = four grandchildren, 3 = eight), the dependent
fetches per tile, wave-wide box and triangle tests, lane fill, the longest tile, and the number of rays whose result
differs from the reference walk (must be 0: the enclosure theorem, see orc_wide_packet_sim)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracedshadows_amd import workloads  # noqa: E402
import oracle  # noqa: E402

_o = oracle._o
_o.orc_wide_packet_sim.restype = None
_o.orc_wide_packet_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_uint32, C.c_void_p, C.c_void_p]
_o.orc_tile_union_stats.restype = None
_o.orc_tile_union_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]


def main():
    args = sys.argv[1:] or ["city", "1920", "1080"]
    for i in range(0, len(args), 3):
        scene, W, H = args[i], int(args[i + 1]), int(args[i + 2])
        wl = workloads.prepare(scene, W, H, light="point", spp=1, via_obj=False, log=print)
        lt = oracle.light_from_product(wl.light, wl.constants)
        k = np.ascontiguousarray(wl.constants.as_array(), np.float32)
        packed = np.ascontiguousarray(wl.packed, np.uint32)
        pos = np.ascontiguousarray(wl.positions, np.float32)
        u = np.zeros(8, np.uint64)
        _o.orc_tile_union_stats(oracle._p(packed), oracle._p(k), C.byref(lt), oracle._p(pos), W, H, oracle._p(u))
        tiles = int(u[0])
        print(f"{scene} {W}x{H}: stackless binary packet: {int(u[1]) / tiles:.1f} steps per tile "
              f"({int(u[4]) / tiles:.1f} of them leaves), {int(u[3]) / tiles / 64:.1f} visits per ray, "
              f"fill {int(u[3]) / max(1, int(u[1])) / 64:.2f}")
        for depth in (1, 2, 3, 12):
            out = np.zeros(16, np.uint64)
            hist = np.zeros(64, np.uint64)
            _o.orc_wide_packet_sim(oracle._p(packed), oracle._p(k), C.byref(lt), oracle._p(pos), W, H, depth, 16,
                                   oracle._p(out), oracle._p(hist))
            t = int(out[0])
            c = np.cumsum(hist) / t
            p99 = int(np.searchsorted(c, 0.99)) * 16
            print(f"  depth {depth}: steps/tile {int(out[1]) / t:7.1f}  box tests/tile {int(out[2]) / t:7.1f} "
                  f"(distinct {int(out[8]) / t:7.1f}, fill {int(out[3]) / max(1, int(out[2])) / 64:.2f})  "
                  f"tri tests/tile {int(out[4]) / t:5.1f} (fill {int(out[5]) / max(1, int(out[4])) / 64:.2f})  "
                  f"longest tile {int(out[6])} steps, p99 ~{p99}  mismatching rays {int(out[7])}  unsafe rays {int(out[10])}"
                  + (f"  cheap test: {int(out[11])} violations, {int(out[12])} false positives" if depth >= 10 else ""))


if __name__ == "__main__":
    main()
