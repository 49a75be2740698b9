"""GPU box: device time (and wall time of the call: upload, build, install) of the GPU builders on the big scenes, no trace, no
host build: `python sah_build_time.py city,courtyard [algo] [repeats]`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracedshadows_amd import api, scenes

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["city", "courtyard"]
algo = sys.argv[2] if len(sys.argv) > 2 else "sah"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
with api.ShadowContext(0) as ctx:
    for name in names:
        sc = scenes.SCENES[name]()
        v, idx = sc.flat()
        ms, wall = [], []
        for _ in range(reps):
            t0 = time.time()
            ms.append(api.bvh_build_device(ctx, v, 8, idx, sc.triangle_count, want_packed=False, install=True, algorithm=algo)[1])
            wall.append((time.time() - t0) * 1e3)
        print(f"{name}: {sc.triangle_count} triangles, {algo} on the device: " + " ".join(f"{m:.2f}" for m in ms) + " ms; wall (upload + build + install): "
              + " ".join(f"{w:.1f}" for w in wall) + " ms", flush=True)
