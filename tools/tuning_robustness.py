"""GPU box: does what rts_ctx_autotune chose on frame A survive a camera or light that moves?  
For a config: tune on camera A (kernel, dissolve threshold, row order, split table), then on cameras advanced 1 % and 5 %
along the view direction (bench.py --scaling weak's path) and on a light swung 90 degrees about the scene's vertical axis,
time (a) the library default (no tuning), (b) A's choice REUSED -- stale table and all --, (c) a fresh tuning on that frame.
Reports the share of the fresh tuning's gain over the default that the stale choice keeps; every timed frame is also
checked against the CPU oracle.
    python tools/tuning_robustness.py atrium_1080p city_4k courtyard_4k"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from raytracedshadows_amd import api, workloads
    import oracle
    for cfg in sys.argv[1:] or ["atrium_1080p"]:
        wl = workloads.prepare_config(cfg, cache=True)
        W, H, sc = wl.W, wl.H, wl.scene
        with api.ShadowContext(0) as ctx:
            ctx.set_bvh(wl.packed)
            d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)

            def frame(advance, swing):
                eye = (sc.eye + (sc.target - sc.eye) * np.float32(advance)).astype(np.float32)
                pos = wl.positions if advance == 0 else api.primary_positions(wl.packed, eye, sc.target, sc.fovy, W, H)[0]
                k = api.RayTracingConstants.make(eye, sc.light_direction, W, H)
                light = wl.light
                if swing:
                    c = (sc.bbox_min + sc.bbox_max) * np.float32(0.5)
                    d = sc.light_point - c
                    light = api.Light.make(api.Light.POINT, np.array([c[0] - d[2], sc.light_point[1], c[2] + d[0]], np.float32))
                return k, pos, light

            def timed(k, light, n=100):
                for _ in range(150):
                    ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                ts = []
                for _ in range(n):
                    ctx.timer_mark(0); ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light); ctx.timer_mark(1)
                    ts.append(ctx.timer_between_ms(0, 1))
                return float(np.median(ts))

            def check(k, pos, light, what):
                want, _, _ = oracle.shadow_mask(wl.packed, k.as_array(), oracle.light_from_product(light, k), pos, W, H)
                ctx.h2d(d_mask, np.full(W * H, 9, np.uint8))
                ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                ctx.synchronize()
                got = np.empty((H, W), np.uint8)
                ctx.d2h(got, d_mask)
                bad = int((got != want).sum())
                if bad:
                    raise SystemExit(f"{cfg} {what}: {bad} mask bytes differ from the oracle")

            def reset():
                ctx.clear_splits()
                for key, v in (("kernel", -1), ("packet_share", 4), ("row_order", 0)):
                    ctx.set_option(key, v)

            def describe():
                return (f"kernel {ctx.get_option('kernel')}, share {ctx.get_option('packet_share')}, rows {ctx.get_option('row_order')}, "
                        f"{ctx.get_option('split_tiles')} split + {ctx.get_option('front_tiles')} front tiles")

            kA, posA, lightA = frame(0.0, False)
            ctx.h2d(d_pos, posA)
            reset()
            default_a = timed(kA, lightA)
            ctx.autotune(kA, d_pos, W, H, d_mask, light=lightA)
            choice = describe()
            tuned_a = timed(kA, lightA)
            check(kA, posA, lightA, "camera A, tuned")
            saved = {key: ctx.get_option(key) for key in ("kernel", "packet_share", "row_order")}
            plan = ctx.split_plan()
            print(f"[{cfg}] camera A: default {default_a:.4f} ms, tuned {tuned_a:.4f} ms ({(tuned_a / default_a - 1) * 100:+.1f} %): {choice}", flush=True)
            steps = (("camera + 0.1 %", 0.001, False), ("camera + 0.3 %", 0.003, False), ("camera + 1 %", 0.01, False), ("camera + 5 %", 0.05, False),
                     ("light swung 90 degrees", 0.0, True))
            seen = {}
            for name, adv, swing in steps:
                k, pos, light = frame(adv, swing)
                # (b) A's choice reused: the options, and the table as it stands (planned on A's frame: never re-measured)
                ctx.h2d(d_pos, posA)
                reset()
                for key, v in saved.items():
                    ctx.set_option(key, v)
                if plan:
                    ctx.plan_splits(kA, d_pos, W, H, d_mask, light=lightA, **{q: plan[q] for q in ("min_life_us", "end_after_us", "piece_us",
                                    "front_life_us", "front_share", "max_pieces", "max_tiles", "xcd_square", "life_block")})
                ctx.h2d(d_pos, pos)
                stale = timed(k, light)
                check(k, pos, light, name + ", A's choice reused")
                reset()
                default = timed(k, light)
                ctx.autotune(k, d_pos, W, H, d_mask, light=light)
                fresh_choice = describe()
                fresh = timed(k, light)
                check(k, pos, light, name + ", tuned afresh")
                gain = default - fresh
                kept = (default - stale) / gain if gain > 0.01 * default else float("nan")
                print(f"[{cfg}] {name}: default {default:.4f} ms, A's choice reused {stale:.4f} ms, tuned afresh {fresh:.4f} ms ({fresh_choice}); "
                      f"reuse keeps {kept * 100:.0f} % of the fresh tuning's gain, loses {(stale / fresh - 1) * 100:+.1f} % against it", flush=True)
                seen[name] = (default, fresh)
            # the same path with the tuner told that the camera will move ("tune_for_motion": tables sorted by blocks of 16 x 16 tiles, no pieces)
            ctx.h2d(d_pos, posA)
            reset()
            ctx.set_option("tune_for_motion", 1)
            ctx.autotune(kA, d_pos, W, H, d_mask, light=lightA)
            ctx.set_option("tune_for_motion", 0)
            motion_a = timed(kA, lightA)
            check(kA, posA, lightA, "camera A, tuned for motion")
            print(f"[{cfg}] camera A, tuned for motion: default {default_a:.4f} ms, tuned {motion_a:.4f} ms ({(motion_a / default_a - 1) * 100:+.1f} %): {describe()}, "
                  f"plan {ctx.split_plan()}", flush=True)
            for name, adv, swing in steps:
                k, pos, light = frame(adv, swing)
                ctx.h2d(d_pos, pos)
                stale = timed(k, light)
                check(k, pos, light, name + ", A's motion-tuned choice reused")
                # ... against the library default measured right beside it (the table switched off, the default options)
                mine = {key: ctx.get_option(key) for key in ("kernel", "packet_share", "row_order")}
                ctx.set_option("tile_splits", 0)
                for key, v in (("kernel", -1), ("packet_share", 4), ("row_order", 0)):
                    ctx.set_option(key, v)
                default = timed(k, light)
                for key, v in mine.items():
                    ctx.set_option(key, v)
                ctx.set_option("tile_splits", 1)
                fresh_gain = 1.0 - seen[name][1] / seen[name][0]
                print(f"[{cfg}] {name}: A's motion-tuned choice reused {stale:.4f} ms against the default beside it {default:.4f} ms: {(stale / default - 1) * 100:+.1f} % "
                      f"({(motion_a / default_a - 1) * 100:+.1f} % on camera A; a fresh tuning of this frame: {-fresh_gain * 100:+.1f} %, "
                      f"the reused motion table keeps {(1 - stale / default) / fresh_gain * 100 if fresh_gain > 0.01 else float('nan'):.0f} % of that)", flush=True)
            ctx.free(d_pos)
            ctx.free(d_mask)


if __name__ == "__main__":
    main()
