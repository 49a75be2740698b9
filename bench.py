#!/usr/bin/env python3
"""Headline benchmark: any-hit shadow rays at 3840x2160 on the ~1M-triangle scene (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config city_4k] [--kernel V] [--scaling strong|weak]

A "step" is one pass of the hot path over one frame of synthetic input: one shadow-mask dispatch
(ray generation + BVH traversal + mask write) with the BVH, the G-buffer positions and the mask
already resident in HBM.  For N > 1 the driver starts one process per GPU
(python -m torch.distributed.run ...); ranks exchange nothing on the data path (BVH replicated
once per GPU, disjoint output), torch.distributed (gloo) is only the barrier / max-over-ranks.

  --scaling strong (default) ONE frame, row-striped over the ranks in interleaved 32-row bands, one dispatch per
                   GPU and step: BASELINE configs[3].  At N = 1 this is the whole frame (configs[2]).
  --scaling weak   every rank traces a full frame of its own (rank r = frame r of a camera path).

Timing: a time-based pre-warm (back-to-back launches until the clocks have ramped), W untimed warm-up steps, then
exactly K steps between barrier + synchronize on both sides; `value` = rays / (max over ranks of that wall time).
HIP events on the launch stream bracket the K launches (`ms_per_frame_gpu_mean`, the roofline's launch duration); right
after the timed region the same launch runs again with an event pair per frame: `ms_per_frame_gpu_median` is the figure
the reference's on-screen counter shows (RayTracedShadows.cpp:263-265).

Rank 0 prints ONE JSON line.  Before timing, every rank checks its GPU mask against the CPU oracle
on every pixel it owns (the correctness gate of SURVEY.md 8d); a mismatch aborts.

`roofline` (N = 1): two bounds the dominant kernel is actually under, each a fraction <= 1 --
  hbm         bytes that crossed the L2's memory side per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc
              passes of THIS command run as child processes before this process touches the GPU; FETCH_SIZE scaled by a
              factor calibrated in the same pass on a frame whose read volume is known) / launch time / 8 TB/s
  valu_issue  wave64 VALU instructions per launch (SQ_INSTS_VALU, same mechanism) / (launch time x measured shader
              clock x 1024 SIMDs) against 0.5 per clock per SIMD (MI355X_MICROARCH.md: 2 cycles per wave64 VALU op)
`bound` names the larger fraction; when that is valu_issue the headline `frac` is its USEFUL share, issue fraction x
lane_fill_members (lanes that hold a ray taking part in the test, counted by the oracle), the raw figure beside it.  The cache-oblivious figure of SURVEY.md 8d (32 V + 16 L + 17 bytes per ray) is
kept as `algorithmic_*`: it exceeds the HBM peak several times because the packet kernel fetches a node once per wave
through the scalar cache, so it prices the shader's memory behaviour, not this kernel's.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the oracle's OpenMP workers must not spin beside the launch loop

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_PER_CLK_SIMD = 0.5   # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32
SIMDS = 256 * 4
BAND = 32

PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"],
              ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"],
              ["TCC_HIT_sum", "TCC_MISS_sum"]]
PMC_PASSES_SECONDARY = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]]
SECONDARY = {"city_4k": ["courtyard_4k", "atrium_1080p", "city_4k_soft16"]}     # driver-run lines besides the headline


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def source_hash():
    """Identifies the kernel build a set of counters belongs to."""
    h = hashlib.sha256()
    for rel in ("rts_kernels.hip", "rts_packet_asm.inc", "rts_wide_asm.inc", "rts_wide.hip", "rts_device.h", "Makefile"):
        h.update(open(os.path.join(ROOT, "raytracedshadows_amd", "csrc", rel), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------
# counters: rocprofv3 child passes of this very command (N = 1 only), run BEFORE this process initialises the GPU
# ---------------------------------------------------------------------------------------------------------------
def _per_dispatch(csv_dir, kernel_substr="shadowMask"):
    """{counter: [value per dispatch, in dispatch order]} summed over the rows rocprofv3 writes per dispatch."""
    acc = {}
    for path in glob.glob(os.path.join(csv_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if kernel_substr not in r.get("Kernel_Name", ""):
                    continue
                key = (r["Counter_Name"], int(r["Dispatch_Id"]))
                acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    out = {}
    for (name, did) in sorted(acc, key=lambda k: k[1]):
        out.setdefault(name, []).append(acc[(name, did)])
    return out


def live_counters(args, say, config=None, kernel=None, passes=None, trace=True, options=None, splits=None):
    """Runs `bench.py --pmc-child` under rocprofv3 once per counter group; returns a dict or None."""
    config = config or args.config
    kernel = args.kernel if kernel is None else kernel
    passes = passes or PMC_PASSES
    prof = shutil.which("rocprofv3")
    if not prof:
        say("rocprofv3 not found: no live counters")
        return None
    work = tempfile.mkdtemp(prefix="rts_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", config, "--kernel", str(kernel),
             "--options", args.options if options is None else options, "--splits", splits_arg(splits), "--prewarm-seconds", "0"]
    res = {"source": "live: rocprofv3 --pmc child passes of this command in this run", "passes": []}
    t_all = time.time()
    try:
        # pass 0: kernel trace (durations under the profiler, grid, registers)
        d = os.path.join(work, "trace")
        r = subprocess.run([prof, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child + ["--prewarm-seconds", "0.5"],
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            say(f"rocprofv3 --kernel-trace failed (rc {r.returncode}): {r.stderr[-300:]}")
            return None
        manifest = json.loads(r.stdout.strip().splitlines()[-1])
        n_cal, n_pre, n_real = manifest["calib_launches"], manifest["prewarm_launches"], manifest["real_launches"]
        n_plan = manifest.get("plan_launches", 0)               # (planning a split table: statistics + planning walk, other instantiations)
        durs = []
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if "shadowMask" in row.get("Kernel_Name", ""):
                        durs.append((int(row["Dispatch_Id"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row))
        durs.sort(key=lambda t: t[0])
        if len(durs) != n_cal + n_plan + n_pre + n_real:
            say(f"kernel trace has {len(durs)} shadow dispatches, expected {n_cal + n_plan + n_pre + n_real}: no live counters")
            return None
        real = [t[1] for t in durs[-n_real:]]
        row = durs[-1][2]
        res["kernel_trace"] = {"kernel": row["Kernel_Name"], "launches": n_real, "avg_ns": sum(real) / len(real),
                               "median_ns": sorted(real)[len(real) // 2],
                               "grid": [int(row.get("Grid_Size_X", 0) or 0), int(row.get("Grid_Size_Y", 0) or 0)],
                               "workgroup": int(row.get("Workgroup_Size_X", 0) or 0),
                               "vgpr": int(row.get("VGPR_Count", 0) or 0), "sgpr": int(row.get("SGPR_Count", 0) or 0)}
        counters, calib = {}, {}
        for i, group in enumerate(passes):
            d = os.path.join(work, f"pmc{i}")
            r = subprocess.run([prof, "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp",
                               env=env, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                say(f"rocprofv3 --pmc {' '.join(group)} failed (rc {r.returncode}): {r.stderr[-300:]}")
                continue
            per = _per_dispatch(d)
            for name, vals in per.items():
                if len(vals) != n_cal + n_plan + n_real:        # (the counter passes run without a pre-warm)
                    continue
                calib[name] = sum(vals[:n_cal]) / n_cal
                counters[name] = sum(vals[-n_real:]) / n_real
            res["passes"].append(group)
        res["counters_per_launch"] = counters
        res["calibration_counters_per_launch"] = calib
        res["calibration"] = manifest["calibration"]
        if manifest.get("split_table"):
            res["split_table_under_the_profiler"] = manifest["split_table"]
        res["seconds"] = round(time.time() - t_all, 1)
        return res if counters else None
    except Exception as e:                                    # a profiler problem must never cost the benchmark line
        say(f"live counters failed: {e!r}")
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def committed_counters(kname, config, say):
    """Fallback: the newest profiles/**/counters_*.json whose kernel source hash, kernel and workload match."""
    want = source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "counters_*.json"), recursive=True), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("source_hash") == want and rec.get("workload") == config and rec.get("kernel") == kname:
            rec["source"] = os.path.relpath(path, ROOT) + " (committed; kernel source hash matches this build)"
            return rec
    say("no committed counter summary matches this kernel build: roofline.traffic = null")
    return None


def roofline_bounds(counters, avg_launch_s, clock_mhz):
    """The two measured bounds of the module docstring from one set of per-launch counters (live passes or the committed
    fallback), the launch time of this run and the shader clock measured in this run.  Pure arithmetic: tested on the CPU
    against the committed counters (tests/test_host_logic.py)."""
    roof = {}
    hbm = issue = None
    if counters:
        c, cal = counters["counters_per_launch"], counters.get("calibration_counters_per_launch", {})
        known = counters.get("calibration", {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # gfx950: FETCH_SIZE reads 1/2 of a wide (16 B/lane) stream (MI355X_MICROARCH.md, HBM); calibrated in the same
            # pass against the frame whose read volume is known rather than assumed
            factor = known["known_read_bytes"] / (cal["FETCH_SIZE"] * 1024) if cal.get("FETCH_SIZE") else 2.0
            fetch = c["FETCH_SIZE"] * 1024 * factor
            write = c["WRITE_SIZE"] * 1024                       # exact for streaming stores; byte stores count 32-B sectors
            traffic = fetch + write
            hbm = {"bytes_per_launch": int(traffic), "fetch_bytes": int(fetch), "write_bytes": int(write),
                   "fetch_factor_calibrated": round(factor, 4),
                   "achieved": round(traffic / avg_launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            hbm["frac"] = round(hbm["achieved"] / HBM_PEAK_GBS, 4)
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                hbm["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "SQ_INSTS_VALU" in c and clock_mhz:
            per = c["SQ_INSTS_VALU"] / (avg_launch_s * clock_mhz * 1e6 * SIMDS)
            issue = {"valu_instr_per_launch": int(c["SQ_INSTS_VALU"]), "achieved": round(per, 4),
                     "peak": VALU_PEAK_PER_CLK_SIMD, "unit": "wave64 VALU instr / clk / SIMD",
                     "frac": round(per / VALU_PEAK_PER_CLK_SIMD, 4)}
            for k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES"):
                if k in c:
                    issue[k.lower()] = int(c[k])
            if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                issue["wait_any_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
            if c.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in c:
                # lanes enabled in EXEC per VALU instruction; the packet loops keep EXEC wide and mask in SGPRs, so this is
                # an upper bound of the useful share -- the algorithmic one is `lane_fill_members` (oracle)
                issue["lane_fill_exec"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3)
        roof["counters_source"] = counters.get("source")
        roof["source_hash"] = source_hash()
        if "kernel_trace" in counters:
            roof["profiler_avg_launch_ms"] = round(counters["kernel_trace"]["avg_ns"] / 1e6, 5)
            roof["kernel_trace"] = counters["kernel_trace"]
    pick = max([b for b in (("hbm", hbm), ("valu_issue", issue)) if b[1]], key=lambda b: b[1]["frac"], default=None)
    if pick:
        roof.update({"bound": pick[0], "achieved": pick[1]["achieved"], "peak": pick[1]["peak"], "unit": pick[1]["unit"],
                     "frac": pick[1]["frac"]})
    else:
        roof.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
    roof["traffic"] = hbm["bytes_per_launch"] if hbm else None
    roof["hbm"] = hbm
    roof["valu_issue"] = issue
    return roof


def headline_fraction(roof, lane_fill_members):
    """the headline `frac` is the USEFUL share of the VALU peak -- issue fraction x the share of lanes that hold a ray
    taking part in the test (counted by the oracle for the same frame) -- so that a kernel which does the same work in fewer
    instructions cannot score lower; the raw issue fraction stays beside it.  Pure arithmetic (tests/test_host_logic.py)."""
    if roof.get("bound") != "valu_issue" or not roof.get("valu_issue") or not lane_fill_members:
        return roof
    raw = roof["valu_issue"]["frac"]
    roof["frac_valu_issue_raw"] = raw
    roof["lane_fill_members"] = lane_fill_members
    roof["frac"] = round(raw * lane_fill_members, 4)
    roof["achieved"] = round(roof["valu_issue"]["achieved"] * lane_fill_members, 4)
    roof["unit"] = "useful wave64 VALU instr / clk / SIMD (issue rate x lanes that hold a participating ray)"
    roof["frac_definition"] = "valu_issue.frac x lane_fill_members"
    return roof


def pmc_child(args):
    """The command the profiler passes run: a few launches on the calibration frame (1-triangle BVH: the read volume is
    the position stream, known), then on the real one.  Prints a one-line manifest."""
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    W, H = wl.W, wl.H
    tri = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]], np.float32)
    one = api.BVHBuilder().build(tri, 3, np.arange(3, dtype=np.uint32), 1).m_packedNodes
    n_cal, n_real = 6, 24
    with api.ShadowContext(0) as ctx:
        if args.kernel >= 0:
            ctx.set_option("kernel", args.kernel)
        apply_options(ctx, args.options)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, wl.positions)
        ctx.set_bvh(one)
        for _ in range(n_cal):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        ctx.set_bvh(wl.packed)
        n_plan, table = 0, None
        if args.splits:                                          # the split table of the tuning child, planned again here
            table = apply_splits(ctx, args.splits, wl, d_pos, d_mask)
            n_plan = 3 if (table and table["split_tiles"]) else 2   # 2 launches with wave statistics (+ the planning walk of the tiles to split)
        n_pre = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.prewarm_seconds:   # only the kernel-trace pass asks for one (durations)
            for _ in range(50):
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            n_pre += 50
        for _ in range(n_real):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        kname = ctx.last_kernel_name()
        ctx.free(d_pos)
        ctx.free(d_mask)
    print(json.dumps({"calib_launches": n_cal, "prewarm_launches": n_pre, "real_launches": n_real, "kernel": kname,
                      "plan_launches": n_plan, "split_table": table,
                      "calibration": {"known_read_bytes": W * H * 16 + 48, "known_write_bytes": W * H,
                                      "what": "same frame and light, BVH of one far-away triangle: reads = the position stream"}}))


# ---------------------------------------------------------------------------------------------------------------
def packet_model(oracle, wl, kname):
    """What the packet walk of this kernel family does on this frame, counted by the oracle (first light sample):
    `lane_fill_members` = rays that take part in a box test / 64 per wave-wide test, and the packet-granular byte model
    (a node is fetched once per WAVE: 32 B per node of a tile's union + 16 B per leaf's v0 for the stackless packet,
    128 B per wide node + 64 B per triangle record for the wide one; + 17 B per pixel for the texel and the mask byte)."""
    import ctypes as C
    o = oracle._o
    W, H = wl.W, wl.H
    k = np.ascontiguousarray(wl.constants.as_array(), np.float32)
    lt = oracle.light_from_product(wl.light, wl.constants)
    packed = np.ascontiguousarray(wl.packed, np.uint32)
    pos = np.ascontiguousarray(wl.positions, np.float32)
    spp = max(1, wl.spp)
    if "wide" in kname:
        o.orc_wide_packet_sim.restype = None
        o.orc_wide_packet_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                          C.c_void_p, C.c_void_p]
        out = np.zeros(16, np.uint64)
        o.orc_wide_packet_sim(oracle._p(packed), oracle._p(k), C.byref(lt), oracle._p(pos), W, H, 12, 16, oracle._p(out), None)
        steps, box_t, box_lanes, tri_t = int(out[1]), int(out[2]), int(out[3]), int(out[4])
        return {"model": "wide packet: 128 B per wide node a tile enters + 64 B per triangle record + 17 B per pixel",
                "nodes_per_tile": round(steps / max(1, int(out[0])), 2), "lane_fill_members": round(box_lanes / max(1, 64 * box_t), 3),
                "bytes_per_launch": int((128 * steps + 64 * tri_t + 17 * W * H) * spp)}
    o.orc_tile_union_stats.restype = None
    o.orc_tile_union_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.zeros(8, np.uint64)
    o.orc_tile_union_stats(oracle._p(packed), oracle._p(k), C.byref(lt), oracle._p(pos), W, H, oracle._p(out))
    U, V, leaf_u = int(out[1]), int(out[3]), int(out[4])
    return {"model": "stackless packet: 32 B per node of a tile's union + 16 B per leaf (v0) + 17 B per pixel",
            "nodes_per_tile": round(U / max(1, int(out[0])), 2), "lane_fill_members": round(V / max(1, 64 * U), 3),
            "bytes_per_launch": int((32 * U + 16 * leaf_u + 17 * W * H) * spp)}


def tune_child(args):
    """`bench.py --tune-child`: which kernel rts_ctx_autotune picks for this workload (its own process, before any profiler pass)."""
    from raytracedshadows_amd import api, workloads
    wl = workloads.prepare_config(args.config, cache=True)
    with api.ShadowContext(0) as ctx:
        ctx.set_bvh(wl.packed)
        d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(wl.W * wl.H)
        ctx.h2d(d_pos, wl.positions)
        for _ in range(100):                                    # ramp the clocks first: the candidates must meet equal conditions
            ctx.trace_shadow_mask_device(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light)
        ctx.synchronize()
        chosen, ms = ctx.autotune(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light)
        tuned = {k: ctx.get_option(k) for k in TUNED_OPTIONS}
        plan = ctx.split_plan()                                 # the split table the third stage kept (None: the plain launch won)
        if plan:
            plan.update(table_size(ctx))
        elif ctx.get_option("tile_order_tiles") and ctx.get_option("tile_order_planned"):
            # (soft shadows: no table, but the same order as a tile order -- rts_ctx_plan_tile_order)
            plan = {"tile_order": {"xcd_square": ctx.get_option("tile_order_square"), "life_block": ctx.get_option("tile_order_block")},
                    "ordered_tiles": ctx.get_option("tile_order_tiles")}
        ctx.free(d_pos)
        ctx.free(d_mask)
    print(json.dumps({"kernel": chosen, "ms": ms, "options": tuned, "splits": plan}))


TUNED_OPTIONS = ("packet_share", "row_order")       # what rts_ctx_autotune sets besides the kernel


def options_arg(options):
    return ",".join(f"{k}={v}" for k, v in sorted((options or {}).items()))


def splits_arg(plan):
    """A split plan (dict of rts_split_plan's numbers) as command-line text for the child processes; '' = no table."""
    if not plan:
        return ""
    if plan.get("tile_order"):
        return f"order:{int(plan['tile_order']['xcd_square'])}:{int(plan['tile_order']['life_block'])}"
    return (":".join(repr(float(plan.get(k, 0.0))) for k in ("min_life_us", "end_after_us", "piece_us", "front_life_us", "front_share"))
            + f":{int(plan['max_pieces'])}:{int(plan.get('max_tiles', 0))}:{int(plan.get('xcd_square', 0))}:{int(plan.get('life_block', 0))}")


def parse_splits(text):
    if not text:
        return None
    f = text.split(":")
    if f[0] == "order":
        return {"tile_order": {"xcd_square": int(f[1]), "life_block": int(f[2]) if len(f) > 2 else 0}}
    return {"min_life_us": float(f[0]), "end_after_us": float(f[1]), "piece_us": float(f[2]), "front_life_us": float(f[3]),
            "front_share": float(f[4]), "max_pieces": int(f[5]), "max_tiles": int(f[6]), "xcd_square": int(f[7]) if len(f) > 7 else 0, "life_block": int(f[8]) if len(f) > 8 else 0}


def apply_splits(ctx, plan, wl, d_pos, d_mask, stripes=None):
    """Plans the table on this process's device with the tuning child's parameters; returns {tiles, pieces} or None."""
    plan = parse_splits(plan) if isinstance(plan, str) else plan
    if not plan:
        return None
    if plan.get("tile_order"):                                   # soft shadows: the tile order instead of a table (2 measuring launches)
        tiles = ctx.plan_tile_order(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light, stripes=stripes,
                                    xcd_square=plan["tile_order"]["xcd_square"], life_block=plan["tile_order"].get("life_block", 0))
        return {"split_tiles": 0, "pieces": 0, "front_tiles": 0, "ordered_tiles": tiles} if tiles else None
    tiles, records = ctx.plan_splits(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light, min_life_us=plan["min_life_us"],
                                     end_after_us=plan["end_after_us"], piece_us=plan["piece_us"], max_pieces=plan["max_pieces"],
                                     front_life_us=plan.get("front_life_us", 0.0), front_share=plan.get("front_share", 0.0),
                                     max_tiles=plan.get("max_tiles", 0), xcd_square=plan.get("xcd_square", 0), life_block=plan.get("life_block", 0), stripes=stripes)
    return table_size(ctx) if tiles else None


def table_size(ctx):
    """What the installed split table holds: tiles split into pieces, their pieces, long tiles merely started first."""
    front = ctx.get_option("front_tiles")
    return {"split_tiles": ctx.get_option("split_tiles"), "pieces": ctx.get_option("split_pieces") - front, "front_tiles": front}


def apply_options(ctx, text):
    for kv in filter(None, (text or "").split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))


def pick_kernel(args, config, say):
    """(kernel id, launch options, split plan) for `config`: the caller's --kernel / --options / --splits, else the autotuner's
    choice (a child process: this one must not touch the GPU before the profiler passes have run)."""
    if args.kernel >= 0:
        return args.kernel, args.options, parse_splits(args.splits)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tune-child", "--config", config], cwd="/tmp",
                           capture_output=True, text=True, timeout=300)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        say(f"autotune [{config}]: kernel {rec['kernel']}, {rec.get('options')}, split table {rec.get('splits')} ({rec['ms']:.4f} ms)")
        return int(rec["kernel"]), options_arg(rec.get("options")), rec.get("splits")
    except Exception as e:
        say(f"autotune child failed for {config} ({e!r}): library default")
        return -1, args.options, None


def measure(ctx, step, steps, warmup, prewarm_seconds, barrier=None, probe_rows=0):
    """The timed region of the contract: time-based pre-warm, W untimed steps, exactly K steps between synchronisations."""
    t0 = time.perf_counter()
    prewarm_launches = 0
    while time.perf_counter() - t0 < prewarm_seconds:
        for _ in range(50):
            step()
        ctx.synchronize()
        prewarm_launches += 50
    ctx.timer_mark(0); ctx.timer_mark(1)                         # (the two events of the timed region exist before it starts)
    for _ in range(warmup):
        step()
    if barrier:
        barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_mark(0)                                            # HIP events on the launch stream around the K launches ...
    for i in range(steps):
        step()
    ctx.timer_mark(1)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    if barrier:
        barrier()
    events_ms = ctx.timer_between_ms(0, 1)                       # ... their span / K = the average launch duration of the timed region
    # per-launch figures (the reference's on-screen counter brackets every frame, cpp:572-594) from a run of the same launch right
    # after the timed region: an event between two launches costs the queue 1-2 us, so the timed region carries none inside
    n_single = max(10, min(steps, 100))
    for i in range(n_single):
        ctx.timer_mark(2 + i)
        step()
    ctx.timer_mark(2 + n_single)
    ctx.synchronize()
    per_launch = np.array([ctx.timer_between_ms(2 + i, 3 + i) for i in range(n_single)])
    clock = None
    if probe_rows:
        # the shader clock from a probed run of the same launch RIGHT AFTER the timed region (the probe -- one wave per tile row
        # stamping both clocks -- is not part of the launches that are timed)
        ctx.set_option("clock_probe", probe_rows)
        ctx.set_option("tile_order", 0)                          # (the probe stamps tile rows of the 2-D grid: a planned tile order is set aside for it)
        try:
            for _ in range(max(10, min(steps, 50))):
                step()
            ctx.synchronize()
            clock = ctx.clock_probe_mhz(probe_rows)
        finally:
            ctx.set_option("clock_probe", 0)
            ctx.set_option("tile_order", 1)
    return {"wall": wall, "kernel_ms": float(events_ms), "median_ms": float(np.median(per_launch)),
            "prewarm_launches": prewarm_launches, "clock_mhz": clock}


def parity_gate(ctx, oracle, wl, rows, step, d_mask, threads, who):
    """GPU mask == CPU oracle mask on every pixel of `rows`; returns (visits, leaf tests, the mask)."""
    W, H = wl.W, wl.H
    olight = oracle.light_from_product(wl.light, wl.constants)
    want = np.zeros((H, W), np.uint8)
    V = L = 0
    for b, e in rows:
        _, v, l = oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, wl.positions, W, H, b, e, threads=threads, out=want)
        V += v
        L += l
    step()
    ctx.synchronize()
    got = np.zeros((H, W), np.uint8)
    ctx.d2h(got, d_mask)
    own = np.zeros(H, bool)
    for b, e in rows:
        own[b:e] = True
    bad = int((got[own] != want[own]).sum())
    if bad:
        raise SystemExit(f"{who}: GPU mask differs from the CPU oracle on {bad} pixels -- not timing")
    return V, L, got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="city_4k")
    ap.add_argument("--kernel", type=int, default=-1, help="kernel variant id (-1 = what rts_ctx_autotune picks for the frame)")
    ap.add_argument("--options", default="", help="context options for the traced frame, key=value,... (with --kernel; else what "
                                                  "rts_ctx_autotune picks: packet_share, row_order)")
    ap.add_argument("--splits", default="", help="split table for the traced frame, min_life_us:end_after_us:piece_us:front_life_us:front_share:max_pieces:max_tiles[:xcd_square[:life_block]], or order:xcd_square:life_block for a planned tile order (with "
                                                 "--kernel; else what rts_ctx_autotune keeps)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--prewarm-seconds", type=float, default=0.6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline workload (no config.secondary)")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the dispatch-floor and shader-clock probes (profiler runs: only the timed kernel is launched)")
    ap.add_argument("--save-counters", default="", help="write the live counter passes to this JSON file (the committed fallback "
                                                          "profiles/**/counters_<config>.json that is used when rocprofv3 is not available)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--tune-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.tune_child:
        return tune_child(args)

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
    say = log if rank == 0 else (lambda *a: None)

    dist = None
    if N > 1:
        import torch.distributed as dist  # control plane only (gloo): barrier + max over ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=N)

    from raytracedshadows_amd import api, partition, scenes, workloads  # raises if librts.so is missing
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    scene_name, W, H, light_kind, spp = workloads.CONFIGS[args.config]
    host_threads = max(1, (os.cpu_count() or 1) // max(1, N))
    striped = args.scaling == "strong" and N > 1

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
    if dist:
        wl = workloads.prepare(scene, W, H, light=light_kind, spp=spp, threads=host_threads, log=say,
                               via_obj=False, packed=shared_packed, radius=workloads.SOFT_RADIUS.get(args.config, 0.01),
                               table=workloads.PER_PIXEL_TABLE.get(args.config, 0))
    else:
        wl = workloads.prepare_config(args.config, cache=True, threads=host_threads, log=say)
    rays_per_frame = wl.rays

    # ---- kernel choice and counters, by child processes while this one has not touched the GPU yet ----------------
    secondary_names = [] if (N > 1 or args.no_secondary) else SECONDARY.get(args.config, [])
    kernel_id, kernel_opts, split_plan = pick_kernel(args, args.config, say) if N == 1 else (args.kernel, args.options, parse_splits(args.splits))
    counters = None
    if N == 1 and not args.no_pmc:
        counters = live_counters(args, say, kernel=kernel_id, options=kernel_opts, splits=split_plan)
    secondary_plan = []
    for name in secondary_names:
        kid, kopts, ksplits = pick_kernel(args, name, say)
        cnt = None if args.no_pmc else live_counters(args, say, config=name, kernel=kid, passes=PMC_PASSES_SECONDARY, options=kopts, splits=ksplits)
        secondary_plan.append((name, kid, cnt, kopts, ksplits))

    my_rows = partition.stripe_rows(H, N, rank, band=BAND, interleaved=True) if striped else [(0, H)]
    my_rays = sum(e - b for b, e in my_rows) * W * max(1, spp)

    # One process per GPU.  RTS_BENCH_SINGLE_DEVICE=1: rehearsal of the multi-rank flow on a one-GPU box (all ranks share
    # device 0); otherwise two ranks on one device would silently halve the result, so that is refused.
    single = bool(os.environ.get("RTS_BENCH_SINGLE_DEVICE"))
    device = 0 if single else local_rank
    if dist:
        import torch
        mine = torch.tensor([device, api.device_count()], dtype=torch.int64)
        seen = [torch.zeros_like(mine) for _ in range(N)]
        dist.all_gather(seen, mine)
        ordinals = [int(t[0]) for t in seen]
        if not single and (len(set(ordinals)) != N or device >= api.device_count()):
            raise SystemExit(f"rank {rank}: device ordinals {ordinals} on a node with {api.device_count()} visible GPUs: one process "
                             "per GPU is required (set RTS_BENCH_SINGLE_DEVICE=1 to rehearse the flow on one device)")
    else:
        ordinals = [device]
    ctx = api.ShadowContext(device)
    ctx.set_bvh(wl.packed)
    d_pos = ctx.malloc(wl.positions.nbytes)
    d_mask = ctx.malloc(W * H)
    ctx.h2d(d_pos, wl.positions)
    ctx.h2d(d_mask, np.zeros((H, W), np.uint8))
    my_stripes = (BAND, N, rank) if striped else None
    split_table = None
    if N > 1 and args.kernel < 0:
        # every rank tunes WHAT IT RUNS, on its own device (untimed): the dispatch of its own stripe -- kernel, dissolve
        # threshold and the split table for its rows (rts_ctx_autotune_stripes); a frame per rank (--scaling weak): the frame
        for _ in range(100):
            if striped:
                ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, BAND, N, rank, light=wl.light)
            else:
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        kernel_id, _ = ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light, stripes=my_stripes)
        kernel_opts = options_arg({k: ctx.get_option(k) for k in TUNED_OPTIONS})
        split_plan = ctx.split_plan()
        if split_plan:
            split_table = table_size(ctx)
        elif ctx.get_option("tile_order_tiles") and ctx.get_option("tile_order_planned"):
            split_plan = {"tile_order": {"xcd_square": ctx.get_option("tile_order_square"), "life_block": ctx.get_option("tile_order_block")}}
            split_table = {"split_tiles": 0, "pieces": 0, "front_tiles": 0, "ordered_tiles": ctx.get_option("tile_order_tiles")}
    else:
        if kernel_id >= 0:
            ctx.set_option("kernel", kernel_id)
        apply_options(ctx, kernel_opts)
        split_table = apply_splits(ctx, split_plan, wl, d_pos, d_mask, stripes=my_stripes)

    def one_step(c=ctx):                  # ONE dispatch per step on every rank
        if striped:
            c.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, BAND, N, rank, light=wl.light)
        else:
            c.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)

    # ---- correctness gate: GPU mask == CPU oracle mask, every pixel this rank owns -----------------
    import oracle  # the checker; never on the measured path
    olight = oracle.light_from_product(wl.light, wl.constants)
    V, L, got = parity_gate(ctx, oracle, wl, my_rows, one_step, d_mask, host_threads, f"rank {rank}")
    say(f"parity gate: {my_rays} rays bit-exact vs oracle (V/ray {V / my_rays:.2f}, L/ray {L / my_rays:.2f})")
    alg_bytes_per_step = 32 * V + 16 * L + 17 * (my_rays // max(1, spp))
    kname = ctx.last_kernel_name()

    # ---- timed region (pre-warm, W warm-up steps, exactly K steps between barrier + synchronize) --------------------
    packet_kernel = kname.startswith("shadowMaskPacketKernel")
    probe_rows = ((H + 7) // 8) if (packet_kernel and not striped and not args.no_probes) else 0
    m = measure(ctx, one_step, args.steps, args.warmup, args.prewarm_seconds, barrier=dist.barrier if dist else None,
                probe_rows=probe_rows)
    wall, kernel_ms, median_ms, clock_mhz, prewarm_launches = m["wall"], m["kernel_ms"], m["median_ms"], m["clock_mhz"], m["prewarm_launches"]

    # ---- extra (not the headline): TWO frames in flight -- consecutive frames on two streams into two masks, so that the
    #      tail of one frame's dispatch (the last waves of a launch keep a few CUs busy) runs beside the next frame's bulk,
    #      as in a renderer that does not wait for frame N before submitting frame N + 1.  Same K steps, same brackets.
    pipelined = None
    if not args.no_probes:
        try:
            s2 = [ctx.stream_create(), ctx.stream_create()]
            d_mask2 = ctx.malloc(W * H)
            ctx.h2d(d_mask2, got)
            masks = [d_mask, d_mask2]

            def step2(i):
                if striped:
                    ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, masks[i & 1], BAND, N, rank, light=wl.light, stream=s2[i & 1])
                else:
                    ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, masks[i & 1], light=wl.light, stream=s2[i & 1])

            for i in range(max(args.warmup, 4)):
                step2(i)
            ctx.synchronize(s2[0]); ctx.synchronize(s2[1])
            if dist:
                dist.barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step2(i)
            ctx.synchronize(s2[0]); ctx.synchronize(s2[1])
            wall2 = time.perf_counter() - t0
            if dist:
                dist.barrier()
            chk = np.zeros((H, W), np.uint8)
            ctx.d2h(chk, d_mask2)
            own = np.zeros(H, bool)
            for b, e in my_rows:
                own[b:e] = True
            pipelined = {"wall": wall2, "parity": bool((chk[own] == got[own]).all())}
            ctx.free(d_mask2)
            ctx.stream_destroy(s2[0]); ctx.stream_destroy(s2[1])
        except Exception as e:
            say(f"two-frames-in-flight measurement failed: {e!r}")

    # ---- the same dispatch against a one-triangle BVH: what the frame costs before any traversal ----------------
    floor_ms = float("nan")
    if not args.no_probes:
        tri = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]], np.float32)
        floor_ctx = api.ShadowContext(device)
        floor_ctx.set_bvh(api.BVHBuilder().build(tri, 3, np.arange(3, dtype=np.uint32), 1).m_packedNodes)
        if kernel_id >= 0:
            floor_ctx.set_option("kernel", kernel_id)
        apply_options(floor_ctx, kernel_opts)
        for _ in range(20):
            one_step(floor_ctx)
        fl = []
        for i in range(30):
            floor_ctx.timer_mark(0)
            one_step(floor_ctx)
            floor_ctx.timer_mark(1)
            fl.append(floor_ctx.timer_between_ms(0, 1))
        floor_ms = float(np.median(fl))
        floor_ctx.close()
        ctx.h2d(d_mask, got)                                        # (the floor frames overwrote the mask)

    wall2 = pipelined["wall"] if pipelined else float("nan")
    if dist:
        import torch
        t2 = torch.tensor([wall2], dtype=torch.float64)
        g2 = [torch.zeros_like(t2) for _ in range(N)]
        dist.all_gather(g2, t2)
        wall2 = max(float(x[0]) for x in g2)
        t = torch.tensor([wall, kernel_ms, median_ms, floor_ms, float(kernel_id), float(ctx.get_option("packet_share")),
                          float(split_table["split_tiles"] if split_table else 0), float(split_table["front_tiles"] if split_table else 0)], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(N)]
        dist.all_gather(gathered, t)
        per_rank = [[float(x) for x in g] for g in gathered]
        wall = max(g[0] for g in per_rank)                          # MAX over ranks
    else:
        per_rank = [[wall, kernel_ms, median_ms, floor_ms, float(kernel_id), 0.0, 0.0, 0.0]]

    frames_per_step = N if (args.scaling == "weak" and N > 1) else 1
    total_rays = rays_per_frame * frames_per_step * args.steps
    value = total_rays / wall / 1e6
    avg_launch_s = kernel_ms / 1e3 / args.steps

    result = {
        "metric": "shadow Mrays/s", "value": round(value, 1), "unit": "Mrays/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {scene_name} ({wl.prim_count} triangles, procedural stand-in), "
                               f"{W}x{H}, 1 {light_kind} light, {max(1, spp)} spp, "
                               + ("one frame per GPU" if frames_per_step > 1 else
                                  f"one frame row-striped over {N} GPUs in interleaved {BAND}-row bands" if striped else "one frame"),
                   "rays_per_frame": rays_per_frame, "kernel": kname,
                   "kernel_choice": "--kernel" if args.kernel >= 0 else ("rts_ctx_autotune on this frame (untimed set-up)" if N == 1 else
                                                                          "rts_ctx_autotune_stripes: every rank on its own stripe's dispatch (untimed set-up)"),
                   "launch_options": kernel_opts,
                   "split_table": dict(split_table, plan=split_plan) if split_table else None,
                   "bvh_bytes": int(wl.packed.nbytes),
                   "ms_per_frame_gpu_median": round(max(g[2] for g in per_rank), 4),
                   "ms_per_frame_gpu_mean": round(max(g[1] for g in per_rank) / args.steps, 4),
                   "dispatch_floor_ms": None if args.no_probes else round(max(g[3] for g in per_rank), 4),
                   "prewarm_launches": prewarm_launches, "device_ordinals": ordinals},
    }
    if pipelined and wall2 == wall2:
        result["config"]["two_frames_in_flight"] = {
            "value": round(rays_per_frame * frames_per_step * args.steps / wall2 / 1e6, 1), "unit": "Mrays/s",
            "ms_per_step": round(wall2 / args.steps * 1e3, 4), "parity": "second mask equal to the first" if pipelined["parity"] else "MISMATCH",
            "note": "not the headline: the same K steps issued alternately on two streams into two masks (consecutive frames "
                    "overlap: one frame's tail beside the next frame's bulk); `value` above is one frame at a time"}
    if N > 1:
        result["config"]["per_rank"] = [{"rank": r, "device": ordinals[r], "wall_ms_per_step": round(g[0] / args.steps * 1e3, 4),
                                         "gpu_median_ms": round(g[2], 4), "dispatch_floor_ms": None if args.no_probes else round(g[3], 4),
                                         "tuned_on_its_own_stripe": {"kernel": int(g[4]), "packet_share": int(g[5]),
                                                                     "split_tiles": int(g[6]), "front_tiles": int(g[7])}}
                                        for r, g in enumerate(per_rank)]

    # ---- roofline (N = 1) -------------------------------------------------------------------------------------------
    roof = {"algorithmic_bytes_per_launch": int(alg_bytes_per_step),
            "algorithmic_GBps": round(alg_bytes_per_step / avg_launch_s / 1e9, 1),
            "avg_launch_ms": round(avg_launch_s * 1e3, 5), "median_launch_ms": round(median_ms, 5), "kernel": kname,
            "shader_clock_mhz": round(clock_mhz, 1) if clock_mhz else None,
            "shader_clock_source": "stamps of 10-50 launches of the same dispatch right after the timed region (first wave of every tile row); "
                                   "the timed launches carry no probe" if clock_mhz else None,
            "note": "frac = the larger of two measured bounds (see the module docstring); algorithmic_* is SURVEY 8d's "
                    "cache-oblivious 32V+16L+17 B/ray from the oracle's exact visit counts, informational; packet_bytes is the same "
                    "model at the granularity this kernel fetches at (once per wave), a fraction of the HBM peak <= 1"}
    if counters and args.save_counters:
        rec = dict(counters, source_hash=source_hash(), workload=args.config, kernel=kname)
        with open(args.save_counters, "w") as fh:
            json.dump(rec, fh, indent=1)
    if N == 1 and counters is None and not args.no_pmc:
        counters = committed_counters(kname, args.config, say)
    roof.update(roofline_bounds(counters, avg_launch_s, clock_mhz))
    if N == 1 and packet_kernel:
        try:
            pm = packet_model(oracle, wl, kname)
            pm["achieved"] = round(pm["bytes_per_launch"] / avg_launch_s / 1e9, 1)
            pm["peak"], pm["unit"] = HBM_PEAK_GBS, "GB/s"
            pm["frac"] = round(pm["achieved"] / HBM_PEAK_GBS, 4)
            roof["packet_bytes"] = pm
            if roof.get("valu_issue"):
                roof["valu_issue"]["lane_fill_members"] = pm["lane_fill_members"]
            headline_fraction(roof, pm["lane_fill_members"])
        except Exception as e:
            say(f"packet model failed: {e!r}")
    result["roofline"] = roof

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

    # ---- the other workloads (N = 1): the same protocol -- parity gate on every pixel, pre-warm, 20 timed steps -- no CPU
    #      baseline; their counters come from their own profiler passes (above) --------------------------------------------
    secondary = {}
    for name, kid, cnt, kopts, ksplits in secondary_plan:
        try:
            t_start = time.time()
            swl = workloads.prepare_config(name, cache=True, threads=host_threads, log=say)
            sW, sH = swl.W, swl.H
            sctx = api.ShadowContext(device)
            sctx.set_bvh(swl.packed)
            if kid >= 0:
                sctx.set_option("kernel", kid)
            apply_options(sctx, kopts)
            sd_pos, sd_mask = sctx.malloc(swl.positions.nbytes), sctx.malloc(sW * sH)
            sctx.h2d(sd_pos, swl.positions)
            stable = apply_splits(sctx, ksplits, swl, sd_pos, sd_mask)

            def sstep():
                sctx.trace_shadow_mask_device(swl.constants, sd_pos, sW, sH, sd_mask, light=swl.light)

            sV, sL, _ = parity_gate(sctx, oracle, swl, [(0, sH)], sstep, sd_mask, host_threads, name)
            sk = sctx.last_kernel_name()
            spacket = sk.startswith("shadowMaskPacketKernel")
            sm = measure(sctx, sstep, 20, 5, args.prewarm_seconds, probe_rows=((sH + 7) // 8) if spacket else 0)
            s_avg = sm["kernel_ms"] / 1e3 / 20
            rec = {"workload": f"{name}: {workloads.CONFIGS[name][0]} ({swl.prim_count} triangles), {sW}x{sH}, {max(1, swl.spp)} spp",
                   "kernel": sk, "launch_options": kopts, "split_table": dict(stable, plan=ksplits) if stable else None, "parity": f"{swl.rays} rays bit-exact vs the oracle", "steps": 20, "warmup": 5,
                   "value": round(swl.rays * 20 / sm["wall"] / 1e6, 1), "unit": "Mrays/s",
                   "ms_per_frame_gpu_median": round(sm["median_ms"], 4),
                   "nodes_per_ray": round(sV / swl.rays, 2), "triangle_tests_per_ray": round(sL / swl.rays, 2),
                   "shader_clock_mhz": round(sm["clock_mhz"], 1) if sm["clock_mhz"] else None}
            sroof = roofline_bounds(cnt, s_avg, sm["clock_mhz"]) if cnt else None
            if sroof:
                rec["bound"], rec["frac"] = sroof["bound"], sroof["frac"]
                rec["valu_issue"] = sroof.get("valu_issue")
                rec["hbm"] = {k: sroof["hbm"][k] for k in ("frac", "achieved", "bytes_per_launch")} if sroof.get("hbm") else None
            if spacket:
                pm = packet_model(oracle, swl, sk)
                pm["frac"] = round(pm["bytes_per_launch"] / s_avg / 1e9 / HBM_PEAK_GBS, 4)
                rec["packet_bytes"] = pm
                if rec.get("valu_issue"):
                    rec["valu_issue"]["lane_fill_members"] = pm["lane_fill_members"]
                    if rec.get("bound") == "valu_issue":
                        rec["frac_valu_issue_raw"], rec["frac"] = rec["frac"], round(rec["frac"] * pm["lane_fill_members"], 4)
            rec["seconds"] = round(time.time() - t_start, 1)
            secondary[name] = rec
            say(f"secondary [{name}]: {rec['value']} Mrays/s, {rec['ms_per_frame_gpu_median']} ms ({sk})")
            sctx.free(sd_pos)
            sctx.free(sd_mask)
            sctx.close()
        except SystemExit:
            raise
        except Exception as e:                                    # a secondary workload must never cost the headline line
            secondary[name] = {"error": repr(e)}
    if secondary:
        result["config"]["secondary"] = secondary

    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()


if __name__ == "__main__":
    main()
