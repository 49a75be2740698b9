#!/usr/bin/env python3
"""Headline benchmark: any-hit shadow rays at 3840x2160 on the ~1M-triangle scene (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config city_4k] [--kernel V] [--scaling weak|strong]

A "step" is one pass of the hot path over one frame of synthetic input: one shadow-mask dispatch
(ray generation + BVH traversal + mask write) with the BVH, the G-buffer positions and the mask
already resident in HBM.  For N > 1 the driver starts one process per GPU
(python -m torch.distributed.run ...); ranks exchange nothing on the data path (BVH replicated
once per GPU, disjoint output), torch.distributed (gloo) is only the barrier / max-over-ranks.

  --scaling weak   (default) every rank traces a full frame of its own (rank r = frame r of a camera
                   path), so per-GPU work is fixed and value = N * rays / time.
  --scaling strong ONE frame, row-striped over the ranks in interleaved 32-row bands, one dispatch per GPU (configs[3]).

Rank 0 prints ONE JSON line.  Before timing, every rank checks its GPU mask against the CPU oracle
on every pixel of its frame (the correctness gate of SURVEY.md 8d); a mismatch aborts.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the oracle's OpenMP workers must not spin beside the launch loop

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_traffic(kernel_name, workload):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/*traffic*.json), or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "*traffic*.json"), recursive=True)):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        for r in rec if isinstance(rec, list) else [rec]:
            if r.get("kernel") == kernel_name and r.get("workload") == workload:
                best = r
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--kernel", type=int, default=-1, help="kernel variant id (-1 = library default)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line (rank 0): everything else any library prints to fd 1 (gloo announces
    # its connections there) goes to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    N = world

    dist = None
    if N > 1:
        import torch.distributed as dist  # control plane only (gloo): barrier + max over ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=N)

    from raytracedshadows_amd import api, partition, scenes, workloads  # raises if librts.so is missing
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    scene_name, W, H, light_kind, spp = workloads.CONFIGS[args.config]
    say = log if rank == 0 else (lambda *a: None)
    host_threads = max(1, (os.cpu_count() or 1) // max(1, N))

    # ---- inputs (untimed): scene -> OBJ -> BVH -> camera -> G-buffer positions -------------------
    # N > 1: the BVH is built ONCE (rank 0) and broadcast as the packed Appendix-A stream (control plane, gloo);
    # every rank uploads it to its own GPU and renders its own G-buffer through it.
    scene = scenes.SCENES[scene_name]()
    if args.scaling == "weak" and N > 1:
        # frame `rank` of a camera path: each GPU renders a different frame of the same scene
        step = (scene.target - scene.eye) * np.float32(0.01 * rank)
        scene.eye = (scene.eye + step).astype(np.float32)
    shared_packed = None
    if dist:
        import torch
        count = int(api.packed_count(scene.triangle_count))
        buf = torch.zeros((count, 4), dtype=torch.int32)
        if rank == 0:
            verts, idx = scene.flat()
            t0 = time.time()
            built = api.BVHBuilder().build(verts, 8, idx, scene.triangle_count).m_packedNodes
            say(f"rank 0 built the BVH in {time.time() - t0:.2f}s; broadcasting {built.nbytes / 1e6:.0f} MB")
            buf.copy_(torch.from_numpy(built.view(np.int32)))
        dist.broadcast(buf, src=0)
        shared_packed = buf.numpy().view(np.uint32)
    wl = workloads.prepare(scene, W, H, light=light_kind, spp=spp, threads=host_threads, log=say,
                           via_obj=not dist, packed=shared_packed)
    rays_per_frame = wl.rays

    BAND = 32
    striped = args.scaling == "strong" and N > 1
    my_rows = partition.stripe_rows(H, N, rank, band=BAND, interleaved=True) if striped else [(0, H)]
    my_rays = sum(e - b for b, e in my_rows) * W * max(1, spp)

    # RTS_BENCH_SINGLE_DEVICE=1: rehearsal of the multi-rank flow on a one-GPU box (all ranks share device 0)
    device = 0 if os.environ.get("RTS_BENCH_SINGLE_DEVICE") else local_rank
    ctx = api.ShadowContext(device)
    ctx.set_bvh(wl.packed)
    if args.kernel >= 0:
        ctx.set_option("kernel", args.kernel)
    d_pos = ctx.malloc(wl.positions.nbytes)
    d_mask = ctx.malloc(W * H)
    ctx.h2d(d_pos, wl.positions)
    ctx.h2d(d_mask, np.zeros((H, W), np.uint8))

    def one_step():                       # ONE dispatch per step on every rank
        if striped:
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, BAND, N, rank, light=wl.light)
        else:
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)

    # ---- correctness gate: GPU mask == CPU oracle mask, every pixel this rank owns -----------------
    import oracle  # the checker; never on the measured path
    olight = oracle.light_from_product(wl.light, wl.constants)
    want = np.zeros((H, W), np.uint8)
    V = L = 0
    for b, e in my_rows:
        _, v, l = oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, wl.positions, W, H, b, e,
                                     threads=host_threads, out=want)
        V += v
        L += l
    one_step()
    ctx.synchronize()
    got = np.zeros((H, W), np.uint8)
    ctx.d2h(got, d_mask)
    mismatches = int((got != want).sum())
    if mismatches:
        raise SystemExit(f"rank {rank}: GPU mask differs from the CPU oracle on {mismatches} pixels -- not timing")
    say(f"parity gate: {my_rays} rays bit-exact vs oracle (V/ray {V / my_rays:.2f}, L/ray {L / my_rays:.2f})")
    alg_bytes_per_step = 32 * V + 16 * L + 17 * (my_rays // max(1, spp))

    # ---- timed region -------------------------------------------------------------------------------
    for _ in range(args.warmup):
        one_step()
    ctx.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(args.steps):
        one_step()
    ctx.timer_end()
    ctx.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ctx.timer_elapsed_ms()          # HIP events on the launch stream, whole timed region
    if dist:
        import torch
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms_max = float(t[0]), float(t[1])
    else:
        kernel_ms_max = kernel_ms

    total_rays = rays_per_frame * (N if args.scaling == "weak" else 1) * args.steps
    value = total_rays / wall / 1e6
    launches = args.steps
    avg_launch_s = kernel_ms / 1e3 / launches
    achieved = alg_bytes_per_step / avg_launch_s / 1e9  # GB/s of algorithmic bytes
    kname = ctx.last_kernel_name()
    # the committed PMC pass measured a whole frame on one GPU; a stripe of a frame is a different launch
    traffic = load_traffic(kname, args.config) if not (args.scaling == "strong" and N > 1) else None

    result = {
        "metric": "shadow Mrays/s", "value": round(value, 1), "unit": "Mrays/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {scene_name} ({wl.prim_count} triangles, procedural stand-in), "
                               f"{W}x{H}, 1 {light_kind} light, {max(1, spp)} spp, "
                               f"{'one frame per GPU' if args.scaling == 'weak' else 'one frame row-striped over GPUs'}",
                   "rays_per_frame": rays_per_frame, "kernel": kname, "bvh_bytes": int(wl.packed.nbytes),
                   "ms_per_frame_gpu_events": round(kernel_ms_max / args.steps, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                     "algorithmic_bytes_per_launch": int(alg_bytes_per_step),
                     "avg_launch_ms": round(avg_launch_s * 1e3, 5), "kernel": kname,
                     "note": "algorithmic (cache-oblivious) bytes 32*V+16*L+17/px from the oracle's exact visit counts; "
                             "frac > 1 means the node stream is served from L2/Infinity Cache, not HBM"},
    }

    # informative extra: instruction-issue view of the same launch.  VALU wave-instructions per launch
    # come from the committed rocprofv3 PMC pass of this kernel + workload, the ceiling from the microbenchmark
    # (profiles/r01/microbench_valu_issue.log); the duration is this run's.
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01", "packet_final_city4k_summary.json")))
        if args.config == "city_4k" and kname == "shadowMaskPacketKernel<1>" and not striped:
            valu = prof["counters_avg_per_dispatch"]["SQ_INSTS_VALU"]
            per_clk = valu / (avg_launch_s * 2.4e9 * 1024)
            result["valu_issue"] = {"achieved": round(per_clk, 3), "peak": 0.325, "unit": "wave64 VALU instr / clk / SIMD (2.4 GHz)",
                                    "frac": round(per_clk / 0.325, 3), "valu_instr_per_launch": int(valu)}
    except Exception:
        pass

    # ---- CPU baseline (rank 0, N == 1 only): the oracle on the host cores, same frame ---------------
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        threads = oracle.max_threads()
        scratch = np.zeros((H, W), np.uint8)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 4.0 or reps < 2:
            oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, wl.positions, W, H, threads=threads, out=scratch)
            reps += 1
        t_all = (time.perf_counter() - t0) / reps
        sample_rows = list(range(16, H, 64))           # every 64th row: the whole frame's mix of cheap and dear rays
        t0 = time.perf_counter()
        for r0 in sample_rows:
            oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, wl.positions, W, H, r0, r0 + 1, threads=1,
                               out=scratch)
        t_one = time.perf_counter() - t0
        rows1 = len(sample_rows)
        result["cpu_baseline"] = {
            "value": round(rays_per_frame / t_all / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"the same full {W}x{H} frame, {reps} repetitions, OpenMP over rows on {threads} threads",
            "value_1thread": round(rows1 * W * max(1, spp) / t_one / 1e6, 3),
            "sample_1thread": f"{rows1} rows (every 64th) of the same frame on 1 thread",
            "bvh_build_seconds": round(wl.build_seconds, 3),
        }

    ctx.free(d_pos)
    ctx.free(d_mask)
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()


if __name__ == "__main__":
    main()
