"""Soak (GPU box): random large frames (packet kernels at full occupancy) on terrain / atrium / cornell / courtyard scenes with random
cameras, lights, sample counts, tuning knobs, stripe layouts and producers of the stream (the host builder or one of the four GPU
builders), each mask against the oracle on the same stream, for a fixed time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from raytracedshadows_amd import api, scenes, workloads

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
cases = 0
tables = 0
with api.ShadowContext(0) as ctx:
    while time.time() - t0 < budget:
        kind = rs.randint(0, 8)
        sc = scenes.terrain(int(rs.choice([40, 90, 160]))) if kind < 3 else (scenes.cornell() if kind < 5 else
                                                                              scenes.SCENES["atrium"]() if kind < 7 else scenes.courtyard())
        verts, idx = sc.flat()
        producer = str(rs.choice(["host", "host", "sah", "ploc", "lbvh", "ploc_sah"]))
        if producer == "host":
            packed = api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes
        else:
            packed, _ = api.bvh_build_device(ctx, verts, 8, idx, sc.triangle_count, algorithm=producer)
        lo, hi = sc.bbox_min, sc.bbox_max
        W, H = int(rs.randint(520, 1500)), int(rs.randint(500, 900))
        eye = (hi + (hi - lo) * rs.random_sample(3) * 0.5 + 1).astype(np.float32)
        if sc.name == "courtyard" and rs.rand() < 0.7:                      # mostly from inside, under the trees
            eye = (lo + (hi - lo) * np.array([0.1 + 0.8 * rs.rand(), 0.12, 0.1 + 0.8 * rs.rand()])).astype(np.float32)
        target = (lo + (hi - lo) * rs.random_sample(3)).astype(np.float32)
        pos, _ = api.primary_positions(packed, eye, target, 1.0, W, H)
        k = api.RayTracingConstants.make(eye, [0.3, 0.8, 0.5], W, H)
        spp = int(rs.choice([1, 1, 1, 3, 16]))
        where = (lo + (hi - lo) * rs.random_sample(3)).astype(np.float32) if rs.rand() < 0.5 else (hi + 5).astype(np.float32)
        table = int(rs.choice([0, 0, 64])) if spp > 1 else 0                 # per-pixel jitter on a third of the soft frames
        light = api.Light.make(api.Light.POINT, where, scenes.jitter_offsets(max(spp, table), 0.5, cases) if spp > 1 else None,
                               nsamples=spp if spp > 1 else None) if rs.rand() < 0.8 else None
        want, _, _ = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(light, k), pos, W, H)
        ctx.set_bvh(packed)
        d_pos, d_mask = ctx.malloc(pos.nbytes), ctx.malloc(W * H)
        ctx.h2d(d_pos, pos)
        try:
            for kernel in (-1, 3, 4, 7, 8, 9):
                ctx.set_option("kernel", kernel)
                ctx.set_option("packet_budget", int(rs.choice([1, 4, 16, 40])))
                ctx.set_option("packet_share", int(rs.choice([0, 2, 4, 9, 16])))
                ctx.set_option("block_waves", int(rs.choice([1, 4])))
                ctx.set_option("xcd_swizzle", int(rs.randint(0, 2)))
                ctx.set_option("row_order", int(rs.randint(0, 3)))
                ctx.set_option("wide_lane", int(rs.randint(0, 2)))
                ctx.set_option("soft_split", int(rs.randint(0, 2)))
                got = np.full((H, W), 7, np.uint8)
                ctx.h2d(d_mask, got)
                n = int(rs.choice([1, 1, 2, 3, 5]))
                if n == 1:
                    ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                else:
                    band = 32 * int(rs.choice([1, 2]))
                    for s in range(n):
                        ctx.trace_shadow_mask_stripes_device(k, d_pos, W, H, d_mask, band, n, s, light=light)
                ctx.synchronize()
                ctx.d2h(got, d_mask)
                bad = int((got != want).sum())
                assert bad == 0, (cases, sc.name, producer, W, H, kernel, spp, n, bad)
                # round 4: a dispatch of several samples once more in a planned tile order (rts_ctx_plan_tile_order)
                if kernel in (3, 8) and spp > 1 and n == 1:
                    ctx.set_option("block_waves", 1); ctx.set_option("xcd_swizzle", 0)
                    ordered = ctx.plan_tile_order(k, d_pos, W, H, d_mask, light=light, xcd_square=int(rs.choice([0, 3, 32])), life_block=int(rs.choice([0, 2, 16])))
                    got = np.full((H, W), 7, np.uint8)
                    ctx.h2d(d_mask, got)
                    ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                    ctx.synchronize()
                    ctx.d2h(got, d_mask)
                    ctx.set_tile_order(None)
                    bad = int((got != want).sum())
                    assert bad == 0, (cases, sc.name, producer, W, H, kernel, spp, "tile order", ordered, bad)
                    tables += 1 if ordered else 0
                # round 4: the same dispatch(es) through a split table with random thresholds, or the tuner's own choice
                if kernel in (3, 8) and spp == 1 and ctx.get_option("wide_nodes") > 0:
                    ctx.set_option("block_waves", 1); ctx.set_option("wide_lane", 0); ctx.set_option("xcd_swizzle", 0)
                    got = np.full((H, W), 7, np.uint8)
                    ctx.h2d(d_mask, got)
                    for s in range(n):
                        stripes = None if n == 1 else (band, n, s)
                        ctx.set_option("kernel", kernel)           # (the tuner installs its own pick, maybe lane-per-ray: no table then)
                        if rs.rand() < 0.25:
                            ctx.autotune(k, d_pos, W, H, d_mask, light=light, stripes=stripes)
                            ctx.h2d(d_mask, got) if s == 0 else None
                        else:
                            plan = dict(min_life_us=float(rs.choice([2.0, 8.0, 30.0, 1e9])),
                                        piece_us=float(rs.choice([1.0, 4.0, 12.0])), max_pieces=int(rs.choice([2, 4, 8, 16])),
                                        end_after_us=float(rs.choice([0.0, 0.0, 20.0])), front_share=float(rs.choice([0.0, 0.03, 0.33, 1.0])),
                                        front_life_us=0.0, max_tiles=int(rs.choice([0, 64, 8192])), xcd_square=int(rs.choice([0, 0, 3, 32])), life_block=int(rs.choice([0, 0, 2, 16])))
                            try:
                                ctx.plan_splits(k, d_pos, W, H, d_mask, light=light, stripes=stripes, **plan)
                            except Exception:
                                print("plan_splits failed:", cases, sc.name, producer, W, H, kernel, stripes, plan, flush=True)
                                raise
                        if n == 1:
                            ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                        else:
                            ctx.trace_shadow_mask_stripes_device(k, d_pos, W, H, d_mask, band, n, s, light=light)
                        tables += 1 if ctx.get_option("split_pieces") else 0
                    ctx.synchronize()
                    ctx.d2h(got, d_mask)
                    ctx.clear_splits()
                    ctx.set_option("kernel", kernel)
                    bad = int((got != want).sum())
                    assert bad == 0, (cases, sc.name, producer, W, H, kernel, "split table", n, bad)
        finally:
            ctx.free(d_pos); ctx.free(d_mask)
            for key, v in (("kernel", -1), ("packet_budget", 16), ("packet_share", 4), ("block_waves", 1), ("xcd_swizzle", 0), ("row_order", 0), ("wide_lane", 1), ("soft_split", 1)):
                ctx.set_option(key, v)
        cases += 1
        if cases % 5 == 0:
            print(f"{cases} big cases ok ({time.time() - t0:.0f}s)", flush=True)
print(f"soak_big: {cases} random large frames x 6 kernels incl. both wide ones (random knobs, stripes, 1-16 samples, per-pixel jitter, host- and GPU-built streams) "
      f"and {tables} dispatches through split tables (random plans or the tuner's) or planned tile orders all bit-exact ({time.time() - t0:.0f}s)")
