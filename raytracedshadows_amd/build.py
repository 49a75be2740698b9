"""Builds the in-tree native libraries (product: librts.so; checker: oracle/librts_oracle.so)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory, *targets):
    subprocess.run(["make", "-C", directory, "-j4", *targets], check=True)


def build_product():
    """hipcc --offload-arch=gfx950 -> raytracedshadows_amd/librts.so (cross-compiles without a GPU)."""
    _make(os.path.join(ROOT, "raytracedshadows_amd", "csrc"))
    return os.path.join(ROOT, "raytracedshadows_amd", "librts.so")


def build_oracle():
    """g++ -> oracle/librts_oracle.so, and oracle/_ref when the reference checkout is present."""
    _make(os.path.join(ROOT, "oracle"))
    if os.path.isdir("/root/reference/External/zeux_objparser"):
        _make(os.path.join(ROOT, "oracle"), "ref")
    return os.path.join(ROOT, "oracle", "librts_oracle.so")
