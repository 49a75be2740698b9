"""CPU: harness and host logic (scenes, camera/G-buffer synthesis, stripe partition)."""
import numpy as np
import pytest

import oracle
from raytracedshadows_amd import api, partition, scenes, workloads


def test_scene_triangle_counts():
    assert scenes.cornell().triangle_count == 1024
    assert scenes.atrium().triangle_count == 249996
    assert scenes.terrain(23).triangle_count == 1058
    assert scenes.courtyard().triangle_count == 999990              # <= 1 000 000: the builder's full-SAH branch


def test_scenes_are_deterministic():
    a, b = scenes.atrium(), scenes.atrium()
    assert (a.verts.view(np.uint32) == b.verts.view(np.uint32)).all() and (a.faces == b.faces).all()


def test_stripes_partition_the_frame():
    for H in (1, 15, 16, 17, 2160, 1080, 131):
        for n in (1, 2, 3, 4, 8):
            for inter in (True, False):
                seen = np.zeros(H, np.int32)
                for r in range(n):
                    for b, e in partition.stripe_rows(H, n, r, 16, inter):
                        assert 0 <= b < e <= H
                        seen[b:e] += 1
                assert (seen == 1).all(), (H, n, inter)
    with pytest.raises(ValueError):
        partition.stripe_rows(10, 2, 2)


def test_primary_positions_land_on_the_geometry():
    wl = workloads.prepare("cornell", 64, 48, via_obj=False)
    pos = wl.positions.reshape(-1, 4)
    hit = pos[:, 3] == 1.0
    assert hit.mean() > 0.5
    world = pos[hit, :3] + wl.scene.eye[None, :]
    lo, hi = wl.scene.bbox_min - 1e-2, wl.scene.bbox_max + 1e-2
    assert ((world >= lo) & (world <= hi)).all()
    assert (pos[~hit] == 0).all()                                   # background = clear value


@pytest.mark.parametrize("scene,W,H", [("cornell", 160, 120), ("atrium", 320, 180), ("terrain", 200, 150)])
def test_gbuffer_pass_equals_the_independent_closest_hit_oracle(scene, W, H):
    """the harness G-buffer (producer of binding 2, contract Model.frag:35-39: camera-relative world
    position, background = clear value) against oracle/'s own closest-hit (explicit-stack descent, written separately):
    positions AND normals bit-for-bit.  On the small scene the oracle's traversal is itself checked against brute force
    over every triangle (no boxes)."""
    wl = workloads.prepare(scenes.terrain(40) if scene == "terrain" else scene, W, H, via_obj=False)
    sc = wl.scene
    pos, nrm, hits = api.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    opos, onrm, ohits = oracle.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    assert hits == ohits and 0 < hits <= W * H
    assert (pos.view(np.uint32) == opos.view(np.uint32)).all()
    assert (nrm.view(np.uint32) == onrm.view(np.uint32)).all()
    if scene != "atrium":
        bpos, bnrm, bhits = oracle.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H, cull=False)
        assert bhits == ohits and (bpos.view(np.uint32) == opos.view(np.uint32)).all()


def test_obj_route_equals_in_memory_route():
    a = workloads.prepare("cornell", 32, 32, via_obj=True)
    b = workloads.prepare("cornell", 32, 32, via_obj=False)
    assert (a.packed == b.packed).all() and (a.positions == b.positions).all()


def test_constants_layout():
    k = api.RayTracingConstants.make([1, 2, 3], [0, 1, 0], 640, 480)
    arr = k.as_array()
    assert arr.shape == (16,) and arr[0:3].tolist() == [1, 2, 3] and arr[8:11].tolist() == [0, 1, 0]
    assert arr[12] == 640 and arr[13] == 480


def test_gbuffer_normals_and_combine(tmp_path):
    wl = workloads.prepare("cornell", 96, 64, via_obj=False)
    pos, nrm, hits = api.primary_gbuffer(wl.packed, wl.scene.eye, wl.scene.target, wl.scene.fovy, 96, 64)
    assert (pos == wl.positions).all()                               # same pass, normals are an extra target
    hit = pos[..., 3] == 1.0
    n = nrm[..., :3]
    assert np.allclose(np.linalg.norm(n[hit], axis=1), 1.0, atol=1e-5) and (n[~hit] == 0).all()
    view = pos[..., :3][hit]                                          # camera-relative position = view vector
    assert ((n[hit] * view).sum(1) <= 1e-4).all()                    # normals face the viewer (Model.frag:38)
    lt = oracle.light_from_product(wl.light, wl.constants)
    mask, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), lt, wl.positions, 96, 64)
    rgb = api.combine(wl.constants, wl.light, pos, nrm, mask)
    assert rgb.shape == (64, 96, 3) and (rgb[~hit] == 0).all() and rgb[hit].min() >= int(0.15 * 255)
    lit = hit & (mask == 1)
    dark = hit & (mask == 0)
    assert rgb[lit].mean() > rgb[dark].mean()                        # Combine.frag: direct light only where lit
    assert rgb[dark].max() <= int(0.2 * 255 + 1)                     # occluded = ambient only (0.15 .. 0.20)
    path = api.write_ppm(str(tmp_path / "c.ppm"), rgb)
    head = open(path, "rb").read(15)
    assert head.startswith(b"P6\n96 64\n255\n")


def test_host_combine_pass_equals_the_oracles_restatement_of_combine_frag():
    """the product's combine pass (host build of the per-pixel code the GPU pass shares) against
    orc_combine, the oracle's own restatement of Combine.frag:18-37 -- point light, 16-sample soft shadows (mask =
    count of unoccluded samples), the reference's directional light from the constants block, and a directional rts_light."""
    wl = workloads.prepare("atrium", 160, 90, via_obj=False)
    W, H, sc = wl.W, wl.H, wl.scene
    pos, nrm, _ = oracle.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    lights = [wl.light, workloads.relight(wl, "point", 16).light, None,
              api.Light.make(api.Light.DIRECTIONAL, [0.3, 0.8, 0.5])]
    for light in lights:
        olight = oracle.light_from_product(light, wl.constants)
        mask, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, pos, W, H)
        got = api.combine(wl.constants, light, pos, nrm, mask)
        want = oracle.combine(wl.constants.as_array(), olight if light is not None else None, pos, nrm, mask)
        assert (got == want).all(), f"{(got != want).any(axis=2).sum()} pixels differ"
        assert want.max() > 100 and len(np.unique(want)) > 8


def test_device_builder_wrapper_rejects_malformed_geometry_before_calling_c():
    """the ctypes view must not hand C a 0-d array or too few indices (C reads 3 * prim_count words)."""
    verts = np.zeros((6, 8), np.float32)
    with pytest.raises(ValueError):
        api.bvh_build_device(None, verts, 8, np.array(7, np.uint32), 2)            # a 0-d array is neither an index array nor a pointer
    with pytest.raises(ValueError):
        api.bvh_build_device(None, verts, 8, np.arange(5, dtype=np.uint32), 2)     # 5 indices for 2 triangles
    with pytest.raises(ValueError):
        api.bvh_build_device(None, (0, 48), 8, np.arange(6, dtype=np.uint32), 2)   # a null device pointer
    with pytest.raises(ValueError):
        api.bvh_build_device(None, np.float32(1.0), 8, np.arange(6, dtype=np.uint32), 2)


def test_bvh_blob_round_trip(tmp_path):
    wl = workloads.prepare("cornell", 8, 8, via_obj=False)
    path = api.save_bvh(str(tmp_path / "c.bvh"), wl.packed)
    back = api.load_bvh(path)
    assert (back == wl.packed).all()
    open(path, "r+b").write(b"XXXX")
    with pytest.raises(api.RtsError):
        api.load_bvh(path)


def _committed(bench):
    """The committed counter summary of the headline workload, whichever packet family the autotuner picked when it was made."""
    for kernel in ("shadowMaskPacketKernel<1,wide>", "shadowMaskPacketKernel<1>"):
        rec = bench.committed_counters(kernel, "city_4k", lambda *a: None)
        if rec is not None:
            return rec
    return None


def test_committed_counter_summary_belongs_to_this_kernel_build():
    """bench.py falls back to profiles/**/counters_<config>.json when rocprofv3 is not available; the file is only used
    if its kernel-source hash matches the sources in the tree.  A kernel change without a fresh profile fails here."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = _committed(bench)
    assert rec is not None, "profiles/**/counters_city_4k.json is missing or belongs to another kernel build: re-run tools/final_evidence.sh"
    c = rec["counters_per_launch"]
    # (129 600 tile waves; a launch with a split table adds the rows of its records -- the tuner keeps a short front list on this frame)
    assert 129600 <= c["SQ_WAVES"] < 129600 * 1.25 and 1.0e8 < c["SQ_INSTS_VALU"] < 2.0e8 and c["FETCH_SIZE"] > 0 and c["WRITE_SIZE"] > 0
    assert 0.3 < c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) <= 1.0          # lanes enabled per VALU instruction


def test_roofline_arithmetic_on_the_committed_counters():
    """bench.py's two bounds recomputed on the CPU from profiles/r02/counters_city_4k.json: fractions of a bound, never above
    1 for any plausible launch time, the larger one named, `traffic` = calibrated fetch + raw write bytes."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = _committed(bench)
    roof = bench.roofline_bounds(rec, 0.1705e-3, 2300.0)
    assert roof["bound"] == "valu_issue" and 0.5 < roof["frac"] < 0.8 and roof["unit"].startswith("wave64 VALU")
    assert 0.3 < roof["valu_issue"]["lane_fill_exec"] <= 1.0
    assert 0.10 < roof["hbm"]["frac"] < 0.20 and abs(roof["hbm"]["fetch_factor_calibrated"] - 2.0) < 0.01
    assert abs(roof["traffic"] - (roof["hbm"]["fetch_bytes"] + roof["hbm"]["write_bytes"])) <= 2 and roof["traffic"] > 150e6
    assert roof["valu_issue"]["frac"] == roof["frac"] and 129600 <= roof["valu_issue"]["sq_waves"] < 129600 * 1.25
    for t in (0.12e-3, 0.17e-3, 0.3e-3):                                   # no launch time this kernel can reach exceeds a bound
        r = bench.roofline_bounds(rec, t, 2400.0)
        assert r["frac"] <= 1.0 and r["hbm"]["frac"] <= 1.0
    none = bench.roofline_bounds(None, 0.17e-3, 2400.0)
    assert none["frac"] is None and none["traffic"] is None and none["bound"] == "hbm"
    no_clock = bench.roofline_bounds(rec, 0.17e-3, None)                   # without a measured clock only the HBM bound is claimed
    assert no_clock["bound"] == "hbm" and no_clock["valu_issue"] is None


def test_bench_passes_the_autotuners_launch_options_on_as_text():
    """bench.py hands what rts_ctx_autotune picked (besides the kernel) to its profiler children and secondary workloads as a
    `key=value,...` string: the round trip sets exactly those options."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = bench.options_arg({"row_order": 1, "packet_share": 6})
    assert text == "packet_share=6,row_order=1"
    assert bench.options_arg({}) == "" and bench.options_arg(None) == ""

    class Recorder:
        def __init__(self):
            self.calls = []

        def set_option(self, k, v):
            self.calls.append((k, v))

    r = Recorder()
    bench.apply_options(r, text)
    assert r.calls == [("packet_share", 6), ("row_order", 1)]
    bench.apply_options(r, "")
    assert len(r.calls) == 2
    assert set(bench.TUNED_OPTIONS) == {"packet_share", "row_order"}


def test_headline_fraction_is_the_useful_share_and_split_plans_round_trip_as_text():
    """`roofline.frac` = issue fraction x lane_fill_members (a kernel that does the same work in fewer
    instructions must not score lower), the raw issue fraction kept beside it; and the split plan of the tuning child reaches
    the profiler children and the parent as text."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    roof = {"bound": "valu_issue", "frac": 0.62, "achieved": 0.31, "unit": "wave64 VALU instr / clk / SIMD",
            "valu_issue": {"frac": 0.62, "achieved": 0.31}}
    out = bench.headline_fraction(dict(roof), 0.79)
    assert out["frac"] == round(0.62 * 0.79, 4) and out["frac_valu_issue_raw"] == 0.62 and out["lane_fill_members"] == 0.79
    assert abs(out["achieved"] - 0.31 * 0.79) < 1e-4 and out["unit"].startswith("useful")
    # fewer instructions at the same time and the same useful work: the raw fraction drops, the useful one does not rise above it
    leaner = bench.headline_fraction({"bound": "valu_issue", "frac": 0.55, "achieved": 0.275, "valu_issue": {"frac": 0.55, "achieved": 0.275}}, 0.89)
    assert abs(leaner["frac"] - out["frac"]) < 0.001
    hbm = {"bound": "hbm", "frac": 0.16, "valu_issue": {"frac": 0.3, "achieved": 0.15}}
    assert bench.headline_fraction(dict(hbm), 0.8)["frac"] == 0.16                 # another bound: untouched
    assert bench.headline_fraction(dict(roof), None)["frac"] == 0.62
    plan = {"min_life_us": 20.25, "end_after_us": 67.5, "piece_us": 10.125, "front_life_us": 0.0, "front_share": 0.3125, "max_pieces": 8,
            "max_tiles": 8192, "xcd_square": 32, "life_block": 16, "split_tiles": 5, "pieces": 20, "front_tiles": 700}
    text = bench.splits_arg(plan)
    back = bench.parse_splits(text)
    assert back == {"min_life_us": 20.25, "end_after_us": 67.5, "piece_us": 10.125, "front_life_us": 0.0, "front_share": 0.3125,
                    "max_pieces": 8, "max_tiles": 8192, "xcd_square": 32, "life_block": 16}
    assert bench.parse_splits("20.25:67.5:10.125:0.0:0.3125:8:8192")["xcd_square"] == 0      # (text of an earlier round's run)
    assert bench.splits_arg(None) == "" and bench.parse_splits("") is None


def _band(us):
    import math
    return math.floor(math.log2(max(us, 0.25)) * 2.0)


def test_split_front_order_is_a_banded_permutation_in_image_order():
    """Host logic of rts_ctx_plan_splits (rtsh_split_front_order): every tile exactly once; half-octave bands of measured life,
    longest first; image order (row-major) inside a band."""
    from raytracedshadows_amd import api
    rs = np.random.RandomState(5)
    bx, by = np.meshgrid(np.arange(61, dtype=np.uint32), np.arange(37, dtype=np.uint32))
    tiles = (bx | (by << np.uint32(16))).reshape(-1)
    life = np.exp(rs.normal(2.0, 1.0, tiles.size)).astype(np.float32)              # 0.3 .. 200 us
    perm = rs.permutation(tiles.size)                                               # (handed in in any order)
    order = api.split_front_order(life[perm], tiles[perm])
    assert sorted(order.tolist()) == list(range(tiles.size))
    t, l = tiles[perm][order], life[perm][order]
    bands = [_band(float(v)) for v in l]
    assert all(a >= b for a, b in zip(bands, bands[1:])) and len(set(bands)) > 8
    key = ((t >> 16).astype(np.int64) << 16) | (t & 0xFFFF)
    for b in set(bands):
        k = key[[i for i, v in enumerate(bands) if v == b]]
        assert (np.diff(k) > 0).all()
    # nothing to order: empty input; the same tile twice is refused
    assert api.split_front_order(np.zeros(0, np.float32), np.zeros(0, np.uint32)).size == 0
    with pytest.raises(api.RtsError):
        api.split_front_order(np.ones(2, np.float32), np.array([7, 7], np.uint32))


def test_split_front_order_deals_image_squares_over_the_xcds():
    """xcd_square S: record first_record + r runs on XCD (first_record + r) mod 8 and is taken from that XCD's S x S-tile squares
    (square (rx, ry) belongs to XCD (rx + 3 ry) mod 8) while it has any left in the band; bands stay where they are and the order
    inside one XCD's tiles of a band stays the image order."""
    from raytracedshadows_amd import api
    S, first = 8, 5
    bx, by = np.meshgrid(np.arange(64, dtype=np.uint32), np.arange(64, dtype=np.uint32))
    tiles = (bx | (by << np.uint32(16))).reshape(-1)
    xcd_of = (((tiles & 0xFFFF) // S) + ((tiles >> 16) // S) * 3) & 7

    # one band, every XCD owns the same number of tiles: every record is one of its XCD's
    order = api.split_front_order(np.full(tiles.size, 10.0, np.float32), tiles, first_record=first, xcd_square=S)
    assert sorted(order.tolist()) == list(range(tiles.size))
    runs_on = (first + np.arange(tiles.size)) & 7
    assert (xcd_of[order] == runs_on).all()
    for x in range(8):                                           # image order inside an XCD's share
        t = tiles[order][runs_on == x]
        k = ((t >> 16).astype(np.int64) << 16) | (t & 0xFFFF)
        assert (np.diff(k) > 0).all()

    # random lives: same bands as without the squares (as multisets per position), most records on their own XCD
    rs = np.random.RandomState(11)
    life = np.exp(rs.normal(2.0, 0.8, tiles.size)).astype(np.float32)
    plain = api.split_front_order(life, tiles)
    dealt = api.split_front_order(life, tiles, first_record=first, xcd_square=S)
    assert sorted(dealt.tolist()) == list(range(tiles.size))
    assert [_band(float(v)) for v in life[plain]] == [_band(float(v)) for v in life[dealt]]
    assert (xcd_of[dealt] == runs_on).mean() > 0.85


def test_split_front_order_by_blocks_only_knows_blocks():
    """life_block B: a tile is as long as the longest tile of its B x B block -- all tiles of a block land in one band, and the
    order does not change when lives move around INSIDE blocks (what a small camera step does)."""
    from raytracedshadows_amd import api
    rs = np.random.RandomState(3)
    B = 4
    bx, by = np.meshgrid(np.arange(40, dtype=np.uint32), np.arange(24, dtype=np.uint32))
    tiles = (bx | (by << np.uint32(16))).reshape(-1)
    block = ((tiles & 0xFFFF) // B) + ((tiles >> 16) // B) * 1000
    life = np.exp(rs.normal(2.0, 1.0, tiles.size)).astype(np.float32)
    order = api.split_front_order(life, tiles, life_block=B, xcd_square=16)
    assert sorted(order.tolist()) == list(range(tiles.size))
    longest = {b: float(life[block == b].max()) for b in np.unique(block)}
    bands = [_band(longest[int(b)]) for b in block[order]]
    assert all(a >= b for a, b in zip(bands, bands[1:]))
    shuffled = life.copy()
    for b in np.unique(block):                                   # the same lives, dealt anew inside every block
        idx = np.flatnonzero(block == b)
        shuffled[idx] = life[idx][rs.permutation(idx.size)]
    assert (api.split_front_order(shuffled, tiles, life_block=B, xcd_square=16) == order).all()
    assert (api.split_front_order(shuffled, tiles, xcd_square=16) != api.split_front_order(life, tiles, xcd_square=16)).any()


def test_bench_carries_a_planned_tile_order_as_text():
    """Soft shadows take no split table; what the tuner keeps for them -- a planned tile order -- travels from bench.py's tuning
    child to its profiler children in the same text channel."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = bench.splits_arg({"tile_order": {"xcd_square": 32, "life_block": 16}, "ordered_tiles": 129600})
    assert text == "order:32:16" and bench.parse_splits(text) == {"tile_order": {"xcd_square": 32, "life_block": 16}}
    assert bench.parse_splits("order:32") == {"tile_order": {"xcd_square": 32, "life_block": 0}}
