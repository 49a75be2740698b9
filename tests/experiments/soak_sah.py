"""Soak (GPU box): the device SAH builder against the oracle's builder with ties by triangle id on random meshes -- soups, coordinates
snapped to coarse grids (equal centroids everywhere), duplicated and degenerate triangles, indexed vertex buffers with strides, lowered
median limits -- byte for byte, for a fixed time."""
import os, resource, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from raytracedshadows_amd import api

resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))   # the oracle recurses as deep as the tree
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t0 = time.time()
cases = 0
sizes = [1, 2, 3, 5, 64, 65, 255, 1023, 1024, 1025, 4097, 20000, 70001, 300000]
with api.ShadowContext(0) as ctx:
    while time.time() - t0 < budget:
        n = int(rs.choice(sizes)) if rs.rand() < 0.7 else int(rs.randint(1, 50000))
        stride = int(rs.choice([3, 4, 8]))
        mode = rs.randint(0, 5)
        if mode in (1, 3):
            n = min(n, 2000)                                            # (equal boxes chain: one level per copy)
        nv = max(3, int(n * rs.choice([0.6, 1.5, 3.0])))
        v = np.zeros((nv, stride), np.float32)
        scale, offset = np.float32(rs.choice([1.0, 100.0, 1e-3])), np.float32(rs.choice([0.0, -50.0, 1e4]))
        if scale < 1 and offset > 1:                                    # two floats per axis: every box is one of a handful
            n, nv = min(n, 3000), min(nv, 9000)
            v = v[:nv]
        v[:, :3] = rs.random_sample((nv, 3)) * scale + offset
        if mode == 1:
            v[:, :3] = np.round(v[:, :3] * 8) / 8                      # coarse grid: ties on every axis
        elif mode == 2:
            v[:, rs.randint(0, 3)] = np.float32(rs.choice([0.0, -0.0, 3.5]))   # flat on one axis (signed zeros)
        idx = rs.randint(0, nv, size=3 * n).astype(np.uint32)
        if mode == 3:
            idx[: 3 * (n // 2)] = np.tile(idx[:3], n // 2)              # half the triangles identical
        if mode == 4 and n > 4:
            idx[3:6] = idx[3]                                           # a point triangle
        limit = int(rs.choice([0, 0, 1, 9, 1000]))
        try:
            got, ms = api.bvh_build_device(ctx, v, stride, idx, n, algorithm="sah", radius=limit)
        except api.RtsError as e:
            raise AssertionError((cases, n, stride, mode, limit, float(scale), float(offset), str(e)))
        want = oracle.bvh_build(v, stride, idx, n, sah_limit=limit or 1000000, ties_by_prim=True)
        assert (got == want).all(), (cases, n, stride, mode, limit, float(scale), float(offset))
        cases += 1
        if cases % 50 == 0:
            print(f"{cases} meshes ok ({time.time() - t0:.0f}s)", flush=True)
print(f"soak_sah: {cases} random meshes (1 .. 300 000 triangles; ties, duplicates, signed zeros, strides, median limits): device SAH stream == "
      f"oracle (ties by triangle id) byte for byte ({time.time() - t0:.0f}s)")
