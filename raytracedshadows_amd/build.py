"""Builds the in-tree product library raytracedshadows_amd/librts.so (hipcc, gfx950)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_product():
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU): HIP kernels + C ABI + host BVH producer."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "raytracedshadows_amd", "csrc"), "-j4"], check=True)
    return os.path.join(ROOT, "raytracedshadows_amd", "librts.so")
