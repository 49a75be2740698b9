"""Soak (GPU box): the randomised parity property of tests/test_gpu_parity.py over many more seeds, for a fixed time.
Every kernel variant x random tuning knobs x four lights per random scene/camera/frame size, each against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from raytracedshadows_amd import api

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
seed = first
with api.ShadowContext(0) as ctx:
    while time.time() - t0 < budget:
        T.test_randomised_scenes_cameras_and_options.__wrapped__(ctx, seed) if hasattr(
            T.test_randomised_scenes_cameras_and_options, "__wrapped__") else T.test_randomised_scenes_cameras_and_options(ctx, seed)
        seed += 1
        if (seed - first) % 20 == 0:
            print(f"{seed - first} random cases ok ({time.time() - t0:.0f}s)", flush=True)
print(f"soak: seeds {first}..{seed - 1} all bit-exact ({seed - first} cases x 5 lights x 10 kernels, random knobs, {time.time() - t0:.0f}s)")
