"""Experiment (GPU box): the team kernel (option "team": waves of a workgroup help each other through LDS)
against the one-wave-per-workgroup stackless packet.  Per point: median / minimum of N single launches between HIP events,
mask bytes that differ from the oracle's, the team kernel's watchdog word.
    TEAMS=0,2,4,8 python tests/experiments/team_ab.py cornell_256 atrium_1080p city_4k courtyard_4k"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracedshadows_amd import api, workloads
import oracle as orc

TEAMS = [int(v) for v in os.environ.get("TEAMS", "0,2,4,8").split(",")]
N = int(os.environ.get("N", "100"))
REPS = int(os.environ.get("REPS", "1"))
TILES = [int(v) for v in os.environ.get("TILES", "4").split(",")]
LOOKS = [int(v) for v in os.environ.get("LOOKS", "1").split(",")]
GIVES = [int(v) for v in os.environ.get("GIVES", "16").split(",")]
for cfg in sys.argv[1:] or ["atrium_1080p"]:
    wl = workloads.prepare_config(cfg, cache=True)
    W, H = wl.W, wl.H
    expect = orc.shadow_mask(wl.packed, wl.constants.as_array(), orc.light_from_product(wl.light, wl.constants), wl.positions, W, H)[0]
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_m = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", 3)
        for rep in range(REPS):
            for team, tiles, look, give in [(t, n, l, g) for t in TEAMS for n in (TILES if t else TILES[:1]) for l in (LOOKS if t else LOOKS[:1])
                                            for g in (GIVES if t else GIVES[:1])]:
                ctx.set_option("team", team)
                ctx.set_option("team_tiles", tiles)
                ctx.set_option("team_look", look)
                ctx.set_option("team_min_give", give)
                ctx.h2d(d_m, np.full(W * H, 7, np.uint8))
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)      # first launch alone: a hang shows here
                ctx.synchronize()
                got = np.empty(W * H, np.uint8)
                ctx.d2h(got, d_m)
                bad = int(np.count_nonzero(got != expect.reshape(-1)))
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    for _ in range(10):
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                    ctx.synchronize()
                ts = []
                for _ in range(N):
                    ctx.timer_mark(0)
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_m, light=wl.light)
                    ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                ctx.synchronize()
                ctx.d2h(got, d_m)
                bad2 = int(np.count_nonzero(got != expect.reshape(-1)))
                print(f"{cfg} team {team} x {tiles} tiles look {look} give {give}: median {np.median(ts):.4f} ms, min {np.min(ts):.4f} = {wl.rays / np.median(ts) / 1e6:.1f} Grays/s; "
                      f"{bad} / {bad2} bytes differ from the oracle (first / last launch); watchdog {ctx.get_option('team_error')} [{ctx.last_kernel_name()}]", flush=True)
        ctx.set_option("team", 0)
