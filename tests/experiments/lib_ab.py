"""Experiment (GPU box): several builds of librts.so on the same box, alternating processes (RTS_LIB selects the build).
usage: python tests/experiments/lib_ab.py <lib A>,<lib B>[,...] [config ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    libs = [os.path.abspath(x) for x in sys.argv[1].split(",")]
    configs = sys.argv[2:] or ["city_4k"]
    res = {}
    for rnd in range(3):
        for lib in libs:
            env = dict(os.environ, RTS_LIB=lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "experiments", "gpu_sweep.py"), "--configs", ",".join(configs),
                                  "--variants", "3", "--warmup", "300", "--frames", "40"], env=env, capture_output=True, text=True).stdout
            for l in out.splitlines():
                if l.startswith("{"):
                    r = json.loads(l)
                    res.setdefault((r["config"], os.path.basename(lib)), []).append((r["ms"], r["mismatch"]))
    for k, v in sorted(res.items()):
        print(k, "ms per round:", [x[0] for x in v], "mismatches:", sum(x[1] for x in v), flush=True)


if __name__ == "__main__":
    main()
