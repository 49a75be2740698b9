"""GPU (MI355X): the HIP path, called through the C ABI of librts.so, against the CPU oracle.

Bar: bit-exact -- every mask byte equal -- for every kernel variant, on every BASELINE.json config at
its full size, on the committed fixtures, under row striping, and on the edge cases of SURVEY.md
Appendix B (axis-parallel rays, NaN/Inf, degenerate triangles, denormals, ragged frame sizes)."""
import os

import numpy as np
import pytest

import oracle
import streams
from raytracedshadows_amd import api, partition, scenes, workloads

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    assert api.device_count() >= 1, "no GPU visible: the shadow path has no CPU fallback"
    c = api.ShadowContext(0)
    yield c
    c.close()


def _variants(ctx):
    return list(range(ctx.get_option("kernel_count")))


def _check_workload(ctx, wl, variants=None, swizzles=(0,)):
    want, V, L = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                    oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
    ctx.set_bvh(wl.packed)
    for sw in swizzles:
        ctx.set_option("xcd_swizzle", sw)
        for v in (variants if variants is not None else _variants(ctx)):
            ctx.set_option("kernel", v)
            got = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
            bad = int((got != want).sum())
            assert bad == 0, f"kernel {v} swizzle {sw}: {bad}/{want.size} bytes differ"
    ctx.set_option("xcd_swizzle", 0)
    return want


def _striped_frames(ctx, wl, want, d_pos, d_mask, counts=(2, 4, 8), name=None, band=32):
    """The frame as `n` row stripes -- contiguous [row_begin, row_end) and interleaved `band`-row bands in ONE dispatch per
    stripe (what rank r of `bench.py --gpus n` launches) -- traced one after the other into one pre-filled mask."""
    W, H = wl.W, wl.H
    for n in counts:
        for interleaved in (False, True):
            got = np.full((H, W), 9, np.uint8)
            ctx.h2d(d_mask, got)
            for r in range(n):
                if interleaved:
                    ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, band, n, r, light=wl.light)
                else:
                    for b, e in partition.stripe_rows(H, n, r, band=band, interleaved=False):
                        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light, row_begin=b, row_end=e)
                if name is not None:
                    assert ctx.last_kernel_name() == name, (ctx.last_kernel_name(), name)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            bad = int((got != want).sum())
            assert bad == 0, f"{name}: {n} stripes, interleaved={interleaved}: {bad} bytes differ"


@pytest.mark.parametrize("name", ["cornell_128", "terrain_96"])
def test_committed_fixtures(ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    H, W = g["mask_dir"].shape
    k = api.RayTracingConstants()
    arr = g["constants"]
    for i in range(4):
        k.cameraPosition[i], k.cameraDirection[i] = arr[i], arr[4 + i]
        k.lightDirection[i], k.renderTargetSize[i] = arr[8 + i], arr[12 + i]
    ctx.set_bvh(g["packed"])
    for v in _variants(ctx):
        ctx.set_option("kernel", v)
        got = ctx.trace_shadow_mask(k, g["positions"], W, H)                     # light=None: the reference path
        assert (got == g["mask_dir"]).all()
        got = ctx.trace_shadow_mask(k, g["positions"], W, H, light=api.Light.make(api.Light.POINT, g["light_point"]))
        assert (got == g["mask_point"]).all()


def test_config0_cornell_256_point_and_directional(ctx):
    wl = workloads.prepare_config("cornell_256")
    _check_workload(ctx, wl, swizzles=(0, 1))
    wl = workloads.prepare("cornell", 256, 256, light="directional")
    _check_workload(ctx, wl)


def test_config1_atrium_1080p_full_size(ctx):
    _check_workload(ctx, workloads.prepare_config("atrium_1080p"), swizzles=(0, 1))


@pytest.fixture(scope="module")
def city4k():
    """The headline workload (BASELINE configs[2]): built once, shared by the configs[2], [3] and [4] tests."""
    wl = workloads.prepare_config("city_4k")
    wl.want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                       oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
    return wl


def test_config2_city_4k_full_size(ctx, city4k):
    want = _check_workload(ctx, city4k, swizzles=(0, 1))
    assert (want == city4k.want).all()
    assert 0.2 < want.mean() < 0.8                                                # a real mix of lit / occluded
    # the library default on a one-sample frame of this size is the wide packet (a static rule: no tuning call needed);
    # the same frame with 16 samples and a 1080p frame keep the stackless packet
    ctx.set_option("kernel", -1)
    got = ctx.trace_shadow_mask(city4k.constants, city4k.positions, city4k.W, city4k.H, light=city4k.light)
    assert ctx.last_kernel_name() == "shadowMaskPacketKernel<1,wide>" and (got == want).all()
    ctx.trace_shadow_mask(city4k.constants, city4k.positions, city4k.W, city4k.H, light=city4k.light, row_begin=0, row_end=1080)
    assert ctx.last_kernel_name() == "shadowMaskPacketKernel<1>"


def test_reference_directional_light_city_4k_full_size(ctx, city4k):
    """The reference's own shader path -- one directional light from the constants block, tmax 1e9 (comp:128-151,
    RayTracedShadows.cpp:245) -- at the headline size: the default kernel, the plain loop and the packet kernels."""
    wl = workloads.relight(city4k, "directional")
    assert wl.light is None                                                       # nothing but RayTracingConstants.lightDirection
    want = _check_workload(ctx, wl, variants=[-1, 0, 3, 4, 7, 8])             # (8: what bench.py times `city_4k_directional` with)
    assert 0.1 < want.mean() < 0.9


def test_config2_courtyard_4k_full_size_hard_scene(ctx):
    """BASELINE configs[2] on the San-Miguel-class stand-in (arcades, nine trees of ~400 000 leaf cards, furniture;
    999 990 triangles): ~60 nodes per ray, rays that scatter between leaves -- packets dissolve, waves run long.
    Every kernel variant at 3840x2160, then 2/4/8 contiguous and interleaved stripes with the default and both wide kernels."""
    wl = workloads.prepare_config("courtyard_4k")
    want = _check_workload(ctx, wl)
    assert 0.2 < want.mean() < 0.8
    W, H = wl.W, wl.H
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        for kernel, name in ((-1, "shadowMaskPacketKernel<1>"), (8, "shadowMaskPacketKernel<1,wide>"),
                             (9, "shadowMaskPacketKernel<1,wide,compiled>")):
            ctx.set_option("kernel", kernel)
            _striped_frames(ctx, wl, want, d_pos, d_mask, name=name)
    finally:
        ctx.set_option("kernel", -1)
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_beyond_the_baseline_sizes_8k_frame_and_median_split_tree(ctx, city4k):
    """Maximum sizes the harness reaches: the 999 488-triangle scene at 7680x4320 (33.2 M rays, 518 400 tiles per dispatch)
    and the 1 034 288-triangle scene -- the builder's > 1 000 000-primitive median-split branch at the top
    (BVHBuilder.cpp:157-178) -- at 3840x2160; default kernel and the plain loop."""
    big = workloads.prepare("city", 7680, 4320, via_obj=False, packed=city4k.packed)
    _check_workload(ctx, big, variants=[-1, 0])
    assert ctx.last_kernel_name() in ("shadowMaskKernel<0>", "shadowMaskPacketKernel<1>")
    del big
    wl = workloads.prepare("city_big", 3840, 2160, via_obj=False)
    assert wl.prim_count > 1000000
    _check_workload(ctx, wl, variants=[-1, 0, 7])


def test_config3_city_4k_row_striped_2_4_8(ctx, city4k):
    """BASELINE configs[3] at its own workload: the 3840x2160 frame of the ~1M-triangle scene cut into 2/4/8 row
    stripes, default kernel (the packet kernel), both partitions the multi-GPU path offers -- contiguous
    [row_begin,row_end) stripes and interleaved 32-row bands in ONE dispatch per stripe -- traced one stripe after
    the other on this device into one mask (RayTracedShadows.comp:128-151 restricted to the owned rows)."""
    wl, W, H = city4k, city4k.W, city4k.H
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        # the default kernel, and the two wide kernels -- 8 is what `bench.py --gpus N` autotunes to on this frame, and a striped
        # dispatch of it runs the BANDS instantiation (rts_kernels.hip: launchShadowMask)
        for kernel, name in ((-1, "shadowMaskPacketKernel<1>"), (8, "shadowMaskPacketKernel<1,wide>"),
                             (9, "shadowMaskPacketKernel<1,wide,compiled>")):
            ctx.set_option("kernel", kernel)
            _striped_frames(ctx, wl, wl.want, d_pos, d_mask, name=name)
        ctx.set_option("kernel", -1)
        # one stripe of an 8-way partition touches only its own rows
        got = np.full((H, W), 9, np.uint8)
        ctx.h2d(d_mask, got)
        (b, e), = partition.stripe_rows(H, 8, 5, band=32, interleaved=False)
        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light, row_begin=b, row_end=e)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        assert (got[b:e] == wl.want[b:e]).all() and (got[:b] == 9).all() and (got[e:] == 9).all()
    finally:
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_config4_city_4k_16_samples_full_size(ctx, city4k):
    """BASELINE configs[4] at its own workload: 3840x2160 x 16 jittered light samples (132.7 M rays) on the ~1M-triangle
    scene with the kernel bench.py uses (AUTO = the SOFT instantiation of the packet kernel) and packet variants 3-5;
    the byte is the number of unoccluded samples."""
    wl = workloads.relight(city4k, "point", 16)
    want = _check_workload(ctx, wl, variants=[-1, 3, 4, 5, 8])
    assert want.max() == 16 and (want == 0).any() and ((want > 0) & (want < 16)).sum() > 10000   # penumbrae exist
    # ... and striped (configs[3] x configs[4]): the wide kernel with 4 waves per tile ("soft_split"), contiguous and interleaved
    W, H = wl.W, wl.H
    ctx.set_option("kernel", 8)
    ctx.set_option("soft_split", 1)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        _striped_frames(ctx, wl, want, d_pos, d_mask, counts=(4, 8), name="shadowMaskPacketKernel<1,wide>")
    finally:
        ctx.free(d_pos)
        ctx.free(d_mask)
        ctx.set_option("kernel", -1)


def test_config4_per_pixel_jitter_every_kernel_and_stripes(ctx):
    """rts_light.table (per-pixel jitter, BASELINE configs[4] as a packet stress): every pixel takes its 16 samples from a
    64-entry table starting at a position hashed from its index.  Every kernel, a ragged frame, stripes (the start depends
    on the pixel's index in the FRAME, not in the stripe); with table == nsamples the start only permutes the samples, so
    the count equals the unjittered one."""
    wl = workloads.prepare("city", 483, 271, spp=16, table=64)
    assert wl.light.table == 64 and wl.light.nsamples == 16
    want = _check_workload(ctx, wl)
    plain = workloads.relight(wl, "point", 16)
    base, _, _ = oracle.shadow_mask(plain.packed, plain.constants.as_array(), oracle.light_from_product(plain.light, plain.constants),
                                    plain.positions, plain.W, plain.H)
    assert (want != base).sum() > 1000                                              # other samples, other penumbra counts
    rot = workloads.relight(wl, "point", 16)
    rot.light.table = 16                                                            # start in the same 16 entries: a permutation
    got = _check_workload(ctx, rot, variants=[-1, 0, 3, 8])
    assert (got == base).all()
    ctx.set_option("kernel", -1)
    for n in (2, 5):
        out = np.full((wl.H, wl.W), 99, np.uint8)
        for r in range(n):
            for b, e in partition.stripe_rows(wl.H, n, r, 16, True):
                ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light, row_begin=b, row_end=e, out=out)
        assert (out == want).all()
    bad = api.Light.make(api.Light.POINT, wl.scene.light_point, scenes.jitter_offsets(8, 1.0), nsamples=16)
    bad.table = 8                                                                   # a table smaller than the sample count
    with pytest.raises(api.RtsError):
        ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=bad)


def test_config4_city_4k_16_samples_per_pixel_jitter_full_size(ctx, city4k):
    """configs[4] as the packet STRESS it is meant to be, at its own size (3840x2160 x 16 samples out of a 64-entry
    table per pixel): the default kernel, the lane-per-ray kernel and both packet families."""
    wl = workloads.relight(city4k, "point", 16, table=64)
    want = _check_workload(ctx, wl, variants=[-1, 7, 3, 8])
    assert want.max() == 16 and ((want > 0) & (want < 16)).sum() > 10000
    ctx.set_option("kernel", -1)


def test_config3_row_stripes_equal_full_frame(ctx):
    """configs[3] on one device: the 2/4/8 stripe sets traced one after the other reproduce the frame."""
    wl = workloads.prepare("atrium", 960, 540)
    want = _check_workload(ctx, wl, variants=[0])
    for n in (2, 4, 8):
        for inter in (True, False):
            out = np.full((wl.H, wl.W), 9, np.uint8)
            for r in range(n):
                for b, e in partition.stripe_rows(wl.H, n, r, 16, inter):
                    ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light,
                                          row_begin=b, row_end=e, out=out)
            assert (out == want).all(), (n, inter)
    out = np.full((wl.H, wl.W), 9, np.uint8)
    ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light, row_begin=100, row_end=100, out=out)
    assert (out == 9).all()                                                        # empty stripe: untouched


def test_config4_soft_shadows_16_samples(ctx):
    wl = workloads.prepare("cornell", 200, 120, spp=16)
    want = _check_workload(ctx, wl)
    assert want.max() == 16 and 0 < (want == 0).sum() and ((want > 0) & (want < 16)).sum() > 0   # penumbra exists
    wl = workloads.prepare("city", 960, 540, spp=16)
    narrow = _check_workload(ctx, wl, variants=[0, 1])
    wide = _check_workload(ctx, workloads.relight(wl, "point", 16, radius=0.05))    # a five times larger light, every kernel
    penumbra = lambda m: ((m > 0) & (m < 16)).mean()
    assert penumbra(wide) > 2 * penumbra(narrow)


@pytest.mark.parametrize("W,H", [(1, 1), (7, 5), (250, 131), (17, 300)])
def test_ragged_frame_sizes(ctx, W, H):
    """The reference has no bounds guard (comp:130, relies on robust access); ours must not touch
    anything outside W x H."""
    wl = workloads.prepare("cornell", W, H, via_obj=False)
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                    oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
    ctx.set_bvh(wl.packed)
    d_pos = ctx.malloc(wl.positions.nbytes)
    d_mask = ctx.malloc(W * H + 256)
    guard = np.full(W * H + 256, 0xAB, np.uint8)
    try:
        for v in _variants(ctx):
            ctx.set_option("kernel", v)
            ctx.h2d(d_pos, wl.positions)
            ctx.h2d(d_mask, guard)
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            got = np.zeros(W * H + 256, np.uint8)
            ctx.d2h(got, d_mask)
            assert (got[:W * H].reshape(H, W) == want).all()
            assert (got[W * H:] == 0xAB).all()
    finally:
        ctx.free(d_pos)
        ctx.free(d_mask)


def _random_rays(n, seed, lo, hi):
    rs = np.random.RandomState(seed)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = lo + rs.random_sample((n, 3)) * (hi - lo)
    d = rs.standard_normal((n, 3))
    rays[:, 4:7] = d
    rays[:, 3] = np.where(rs.random_sample(n) < 0.5, 1e9, rs.random_sample(n) * 20)
    return rays


def test_generic_rays_including_nan_inf_and_axis_parallel(ctx):
    sc = scenes.cornell()
    verts, idx = sc.flat()
    packed = api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes
    rays = _random_rays(200000, 3, sc.bbox_min - 2, sc.bbox_max + 2)
    n = rays.shape[0]
    # axis-parallel directions: exact zeros -> 1/0 = inf -> 0*inf = NaN on slab planes (Appendix B-3)
    rays[0:40000, 4] = 0.0
    rays[20000:60000, 5] = 0.0
    rays[50000:70000, 6] = 0.0
    # origins exactly on bounding planes and grid lines of the scene (walls at 0 and 10, grid step 1.25)
    rays[0:70000, 0:3] = np.round(rays[0:70000, 0:3] / 1.25) * 1.25
    rays[70000:70100, 0] = np.nan
    rays[70100:70200, 5] = np.inf
    rays[70200:70300, 1] = -np.inf
    rays[70300:70400, 4:7] = 0.0                                                   # null direction
    rays[70400:70500, 3] = np.nan                                                  # NaN tmax
    rays[70500:70600, 4] = 1e-42                                                   # denormal direction -> 1/d = inf
    rays[70600:70700, 3] = 0.0
    want, _, _ = oracle.trace_rays(packed, rays)
    ctx.set_bvh(packed)
    for v in _variants(ctx):
        ctx.set_option("kernel", v)
        got = ctx.trace_rays(rays)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"kernel {v}: {bad.size} of {n} rays differ, first {bad[:5]}"
    assert 0 < want.sum() < n
    # device-pointer entry, default variant (generic rays: lane-per-ray with work sharing)
    ctx.set_option("kernel", -1)
    d_rays, d_out = ctx.malloc(rays.nbytes), ctx.malloc(n)
    try:
        ctx.h2d(d_rays, rays)
        ctx.trace_rays_device(d_rays, n, d_out)
        ctx.synchronize()
        got = np.zeros(n, np.uint8)
        ctx.d2h(got, d_out)
        assert (got == want).all()
        assert ctx.last_kernel_name() == "traceRaysKernel<7>"
    finally:
        ctx.free(d_rays)
        ctx.free(d_out)


def test_degenerate_and_denormal_triangles(ctx):
    """Zero-area triangles 'hit' through the all-NaN rule (Appendix B-4); denormal edges must not be flushed."""
    rs = np.random.RandomState(5)
    P = 400
    c = rs.random_sample((P, 1, 3)) * 10
    tri = (c + (rs.random_sample((P, 3, 3)) - 0.5)).astype(np.float32)
    tri[0:50, 1] = tri[0:50, 0]                                                    # v1 == v0
    tri[50:100, 2] = tri[50:100, 1]                                                # v2 == v1 (sliver to a line)
    tri[100:150] = tri[100:150, 0:1]                                               # all three equal (a point)
    tiny = np.float32(1e-41)
    tri[150:200, 1] = tri[150:200, 0] + np.array([tiny, 0, 0], np.float32)         # denormal-ish edge lengths
    base = np.zeros((50, 3, 3), np.float32)
    base[:, 1, 0] = tiny * 3
    base[:, 2, 1] = tiny * 5
    tri[200:250] = base                                                            # denormal triangles at the origin
    verts = tri.reshape(-1, 3)
    idx = np.arange(3 * P, dtype=np.uint32)
    packed = api.BVHBuilder().build(verts, 3, idx, P).m_packedNodes
    assert (packed == oracle.bvh_build(verts, 3, idx, P)).all()
    rays = _random_rays(60000, 8, np.float32(-1), np.float32(11))
    rays[:20000, 0:3] = verts[rs.randint(0, verts.shape[0], 20000)] - rays[:20000, 4:7] * 2   # aim at vertices
    rays[20000:22000, 0:3] = 0.0
    rays[20000:22000, 4:7] *= 1e-3
    want, _, _ = oracle.trace_rays(packed, rays)
    ctx.set_bvh(packed)
    for v in _variants(ctx):
        ctx.set_option("kernel", v)
        got = ctx.trace_rays(rays)
        assert (got == want).all(), f"kernel {v}: {(got != want).sum()} rays differ"


def _deep_bushy_stream(levels):
    """A tree whose wide walk keeps three subtrees pending per level: node(l) = ((node(l-1), small), (small, small)), small =
    two tiny triangles in opposite corners of the unit square (its box covers the middle, where every ray passes; the
    triangles are hit by nobody).  Level 0 holds two real occluders."""
    tris, z = [], [0.0]

    def tiny(x, y):
        tris.append([[x, y, z[0]], [x + 1e-3, y, z[0]], [x, y + 1e-3, z[0]]])
        return len(tris) - 1

    def small():
        z[0] += 0.25
        a = tiny(0.05, 0.05)
        z[0] += 0.25
        return (a, tiny(0.95, 0.95))

    def node(l):
        if l == 0:
            z[0] += 0.5
            tris.append([[0.0, 0.0, z[0]], [0.55, 0.0, z[0]], [0.0, 0.9, z[0]]])          # occludes the rays of one corner
            tris.append([[0.6, 0.6, z[0] + 0.1], [0.9, 0.6, z[0] + 0.1], [0.6, 0.9, z[0] + 0.1]])
            return (len(tris) - 2, len(tris) - 1)
        return ((node(l - 1), small()), (small(), small()))

    tree = node(levels)
    return streams.stream_from_tree(tree, np.array(tris, np.float32))


def test_wide_stack_limit_dissolves_and_keeps_every_pixel(ctx):
    """the wide packet's stack lives in lanes 0..62 of three VGPRs and lane 63 keeps EXEC of the entry; a walk
    that pushes three subtrees per level for 25 levels must dissolve BEFORE entry 63 is written, and every pixel of the tile
    must still be traversed and stored.  packet_share = 0 switches the coherence rule off, so the stack limit is the only way
    to dissolve -- the wave statistics must show that it did."""
    packed = _deep_bushy_stream(25)
    assert api.bvh_validate(packed) == (packed.shape[0] + 2) // 5
    W = H = 24
    pos = np.zeros((H, W, 4), np.float32)
    pos[..., 0] = 0.2 + 0.6 * (np.arange(W, dtype=np.float32)[None, :] + 0.5) / W
    pos[..., 1] = 0.2 + 0.6 * (np.arange(H, dtype=np.float32)[:, None] + 0.5) / H
    k = api.RayTracingConstants.make([0, 0, 0], [0.001, 0.002, 1.0], W, H)
    want, _, _ = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(None, k), pos, W, H)
    assert 0 < want.sum() < want.size
    ctx.set_bvh(packed)
    assert ctx.get_option("wide_nodes") > 25
    waves = (W // 8) * (H // 8)
    d_pos, d_mask = ctx.malloc(pos.nbytes), ctx.malloc(W * H)
    ctx.h2d(d_pos, pos)

    def trace():                                    # device pointers and a pre-filled mask: a pixel nobody stores shows
        got = np.full((H, W), 9, np.uint8)
        ctx.h2d(d_mask, got)
        ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        return got

    try:
        ctx.set_option("packet_share", 0)
        for kernel in (8, 9):
            for lane in (0, 1):
                ctx.set_option("kernel", kernel)
                ctx.set_option("wide_lane", lane)
                got = trace()
                assert (got == want).all(), (kernel, lane, int((got != want).sum()))
                ctx.set_option("wave_stats", waves)
                got = trace()
                st = ctx.read_wave_stats(waves)
                ctx.set_option("wave_stats", 0)
                assert (got == want).all(), (kernel, lane, "with wave statistics")
                assert (st[:, 2] & 1).all(), "every tile's packet must have run into the stack limit and dissolved"
        ctx.set_option("packet_share", 4)
        for kernel in _variants(ctx):                                   # and the stream as such, every kernel
            ctx.set_option("kernel", kernel)
            assert (trace() == want).all(), kernel
    finally:
        for key, v in (("kernel", -1), ("packet_share", 4), ("wide_lane", 0), ("wave_stats", 0)):
            ctx.set_option(key, v)
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_stream_with_orphan_nodes_gets_no_private_copy(ctx):
    """rts_bvh_validate accepts a stream in which a leaf's miss link skips nodes (the skipped nodes are
    orphans no walk ever reaches).  The device-side verdict must then say "not a pre-order tree" so that no wide copy is
    derived from parent entries nobody wrote; every kernel still equals the oracle on that very stream."""
    tris = np.zeros((3, 3, 3), np.float32)
    tris[0] = [[0, 0, 5], [4, 0, 5], [0, 4, 5]]
    tris[1] = [[6, 6, 7], [9, 6, 7], [6, 9, 7]]
    tris[2] = [[2, 2, 3], [3, 2, 3], [2, 3, 3]]                       # only the stray leaves point at it
    good = streams.stream_from_tree(((0, 1), 2), tris)                # N = 5: 0 inner, 1 inner, 2 leaf, 3 leaf, 4 leaf
    N = 5
    bad = good.copy()
    # root 0 -> children: leaf 1 (link 4) and leaf 4; nodes 2 and 3 are stray leaves
    f = bad.view(np.float32)
    f[2, :3] = tris[0, 1] - tris[0, 0]; bad[2, 3] = 2 * N + 0
    f[3, :3] = tris[0, 2] - tris[0, 0]; bad[3, 3] = 4
    for i in (2, 3):
        f[2 * i, :3] = tris[2, 1] - tris[2, 0]; bad[2 * i, 3] = 2 * N + 2
        f[2 * i + 1, :3] = tris[2, 2] - tris[2, 0]; bad[2 * i + 1, 3] = i + 1
    f[8, :3] = tris[1, 1] - tris[1, 0]; bad[8, 3] = 2 * N + 1
    f[9, :3] = tris[1, 2] - tris[1, 0]; bad[9, 3] = 0xFFFFFFFF
    assert api.bvh_validate(bad) == 3
    W, H = 64, 64
    pos = np.zeros((H, W, 4), np.float32)
    pos[..., 0] = 10.0 * (np.arange(W, dtype=np.float32)[None, :] + 0.5) / W
    pos[..., 1] = 10.0 * (np.arange(H, dtype=np.float32)[:, None] + 0.5) / H
    k = api.RayTracingConstants.make([0, 0, 0], [0.001, 0.002, 1.0], W, H)
    for blob, enclosed in ((good, 1), (bad, 0)):
        want, _, _ = oracle.shadow_mask(blob, k.as_array(), oracle.light_from_product(None, k), pos, W, H)
        assert 0 < want.sum() < want.size
        ctx.set_bvh(blob)
        assert ctx.get_option("bvh_enclosed") == enclosed
        assert (ctx.get_option("wide_nodes") > 0) == bool(enclosed)
        for v in _variants(ctx):
            ctx.set_option("kernel", v)
            assert (ctx.trace_shadow_mask(k, pos, W, H) == want).all(), (enclosed, v)
    ctx.set_option("kernel", -1)


def _device_frame(ctx, wl, d_pos, d_mask, **kw):
    got = np.full((wl.H, wl.W), 9, np.uint8)
    ctx.h2d(d_mask, got)
    ctx.trace_shadow_mask_device(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light, **kw)
    ctx.synchronize()
    ctx.d2h(got, d_mask)
    return got


def test_split_tables_never_change_the_mask(ctx):
    """rts_ctx_plan_splits: tiles measured to be long are walked by several one-wave pieces, each over one index range of the
    stream.  Whatever the table -- few tiles or thousands, 2 or 16 pieces, planned from this frame, from another camera's
    frame or from statistics handed in -- the mask is the oracle's; the table is only used by the dispatch it was planned for,
    and dropped with the stream."""
    wl = workloads.prepare("atrium", 960, 540)
    W, H = wl.W, wl.H
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        for kernel in (3, 8):
            ctx.set_option("kernel", kernel)
            # (life, piece, max pieces, ended after, front tiles: lived longer than / the longest share)
            # (the last number: xcd_square -- the front order dealt over the XCDs by image squares)
            for life, piece, maxp, end, front, share, square in (
                    (40.0, 10.0, 8, 0.0, 0.0, 0.0, 0), (10.0, 3.0, 16, 0.0, 0.0, 0.0, 0), (3.0, 1.0, 4, 0.0, 0.0, 0.0, 5),
                    (30.0, 8.0, 8, 0.0, 6.0, 0.0, 0), (1e9, 1e9, 8, 0.0, 0.0, 0.5, 16), (25.0, 8.0, 8, 20.0, 0.0, 1.0, 0),
                    (1e9, 1e9, 8, 0.0, 0.0, 1.0, 32), (25.0, 8.0, 8, 20.0, 0.0, 1.0, 1), (15.0, 5.0, 8, 40.0, 0.0, 0.0, 0)):
                tiles, records = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=life, piece_us=piece,
                                                 max_pieces=maxp, end_after_us=end, front_life_us=front, front_share=share, xcd_square=square,
                                                 life_block=square // 2)
                assert ctx.split_plan()["xcd_square"] == square and ctx.split_plan()["life_block"] == square // 2
                split, fronts = ctx.get_option("split_tiles"), ctx.get_option("front_tiles")
                assert split + fronts == tiles and ctx.get_option("split_pieces") == records and records >= 2 * split + fronts
                assert (fronts > 100) == (front > 0 or share > 0 or split == 4096), (life, front, share, split, fronts)
                got = _device_frame(ctx, wl, d_pos, d_mask)
                assert (got == want).all(), (kernel, life, piece, maxp, front, share, tiles, int((got != want).sum()))
            assert tiles > 100                                            # the aggressive plans really split many tiles
            plan = ctx.split_plan()
            assert plan is not None and plan["max_pieces"] == 8 and abs(plan["min_life_us"] - 15.0) < 1e-6
            # the table is for the full-frame dispatch only: a row range runs without it, and is right
            got = np.full((H, W), 9, np.uint8)
            ctx.h2d(d_mask, got)
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light, row_begin=64, row_end=200)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            assert (got[64:200] == want[64:200]).all() and (got[:64] == 9).all() and (got[200:] == 9).all()
            # ... unless the table was planned FOR that row range (what a rank with a contiguous stripe does)
            tiles_rr, _ = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=15.0, piece_us=5.0, max_pieces=8,
                                          front_share=1.0, row_begin=64, row_end=200)
            assert tiles_rr == ((W + 7) // 8) * ((200 - 64) // 8)
            got = np.full((H, W), 9, np.uint8)
            ctx.h2d(d_mask, got)
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light, row_begin=64, row_end=200)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            assert (got[64:200] == want[64:200]).all() and (got[:64] == 9).all() and (got[200:] == 9).all()
            assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all()          # (the full frame: another dispatch, no table)
            tiles, pieces = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=15.0, piece_us=5.0, max_pieces=8,
                                            end_after_us=40.0)
            # "tile_splits" 0 ignores the table
            ctx.set_option("tile_splits", 0)
            assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all()
            ctx.set_option("tile_splits", 1)
            # another camera (the G-buffer of a frame 3 % further along the view direction) through the SAME table
            sc = wl.scene
            eye2 = (sc.eye + (sc.target - sc.eye) * np.float32(0.03)).astype(np.float32)
            pos2, _ = api.primary_positions(wl.packed, eye2, sc.target, sc.fovy, W, H)
            k2 = api.RayTracingConstants.make(eye2, [0.3, 0.8, 0.5], W, H)
            want2, _, _ = oracle.shadow_mask(wl.packed, k2.as_array(), oracle.light_from_product(wl.light, k2), pos2, W, H)
            d_pos2 = ctx.malloc(pos2.nbytes)
            ctx.h2d(d_pos2, pos2)
            got = np.full((H, W), 9, np.uint8)
            ctx.h2d(d_mask, got)
            ctx.trace_shadow_mask_device(k2, d_pos2, W, H, d_mask, light=wl.light)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            ctx.free(d_pos2)
            assert ctx.get_option("split_tiles") > 0 and (got == want2).all(), (kernel, "moved camera", int((got != want2).sum()))
            # statistics of an earlier frame handed in instead of a measuring launch
            waves = ((W + 7) // 8) * ((H + 7) // 8)
            ctx.clear_splits()
            ctx.set_option("wave_stats", waves)
            _device_frame(ctx, wl, d_pos, d_mask)
            st, rt = ctx.read_wave_stats(waves), ctx.read_wave_realtime(waves)
            ctx.set_option("wave_stats", 0)
            tiles, pieces = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=20.0, piece_us=5.0, max_pieces=8,
                                            prev=(st, rt))
            assert tiles > 0
            assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all()
            # two frames in flight on two streams share the table, not its state
            s0, s1 = ctx.stream_create(), ctx.stream_create()
            d_mask2 = ctx.malloc(W * H)
            for rep in range(3):
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light, stream=s0)
                ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask2, light=wl.light, stream=s1)
            ctx.synchronize(s0); ctx.synchronize(s1)
            for d in (d_mask, d_mask2):
                got = np.zeros((H, W), np.uint8)
                ctx.d2h(got, d)
                assert (got == want).all()
            ctx.free(d_mask2)
            ctx.stream_destroy(s0); ctx.stream_destroy(s1)
        # a new stream drops the table
        ctx.set_bvh(wl.packed)
        assert ctx.get_option("split_tiles") == 0 and ctx.split_plan() is None
        # interleaved stripes: a table per stripe dispatch
        ctx.set_option("kernel", 8)
        got = np.full((H, W), 9, np.uint8)
        ctx.h2d(d_mask, got)
        for r in range(3):
            tiles, _ = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=8.0, piece_us=3.0, max_pieces=8,
                                       stripes=(32, 3, r))
            assert tiles > 0
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, 3, r, light=wl.light)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        assert (got == want).all(), int((got != want).sum())
    finally:
        ctx.clear_splits()
        ctx.set_option("kernel", -1)
        ctx.set_option("tile_splits", 1)
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_split_tables_on_awkward_streams_and_rays(ctx):
    """Pieces on the hand-made deep tree (their stack limit and the lane-per-ray continuation inside an index range), with an
    axis-parallel light (every wave takes the exact walk: piece 0 walks it whole, the others contribute nothing), on a ragged
    frame, and with soft shadows (no table is planned for them)."""
    packed = _deep_bushy_stream(25)
    W = H = 24
    pos = np.zeros((H, W, 4), np.float32)
    pos[..., 0] = 0.2 + 0.6 * (np.arange(W, dtype=np.float32)[None, :] + 0.5) / W
    pos[..., 1] = 0.2 + 0.6 * (np.arange(H, dtype=np.float32)[:, None] + 0.5) / H
    ctx.set_bvh(packed)
    d_pos, d_mask = ctx.malloc(pos.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, pos)
        for direction in ([0.001, 0.002, 1.0], [0.0, 0.0, 1.0]):
            k = api.RayTracingConstants.make([0, 0, 0], direction, W, H)
            want, _, _ = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(None, k), pos, W, H)
            for kernel in (3, 8):
                for share in (4, 0):
                    ctx.set_option("kernel", kernel)
                    ctx.set_option("packet_share", share)
                    tiles, pieces = ctx.plan_splits(k, d_pos, W, H, d_mask, min_life_us=0.5, piece_us=0.5, max_pieces=16)
                    assert tiles == 9 and pieces > 18
                    got = np.full((H, W), 9, np.uint8)
                    ctx.h2d(d_mask, got)
                    ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask)
                    ctx.synchronize()
                    ctx.d2h(got, d_mask)
                    assert (got == want).all(), (direction, kernel, share, int((got != want).sum()))
    finally:
        ctx.clear_splits()
        ctx.set_option("kernel", -1)
        ctx.set_option("packet_share", 4)
        ctx.free(d_pos)
        ctx.free(d_mask)
    wl = workloads.prepare("cornell", 250, 131, via_obj=False)
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(wl.W * wl.H)
    try:
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", 3)
        tiles, _ = ctx.plan_splits(wl.constants, d_pos, wl.W, wl.H, d_mask, light=wl.light, min_life_us=2.0, piece_us=1.0, max_pieces=8)
        assert tiles > 0
        assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all()
        soft = workloads.relight(wl, "point", 16)
        with pytest.raises(api.RtsError):
            ctx.plan_splits(soft.constants, d_pos, wl.W, wl.H, d_mask, light=soft.light)
    finally:
        ctx.clear_splits()
        ctx.set_option("kernel", -1)
        ctx.free(d_pos)
        ctx.free(d_mask)


@pytest.mark.parametrize("name", ["atrium_1080p", "city_4k", "courtyard_4k"])
def test_split_tables_full_size(ctx, name):
    """The BASELINE frames at their own size through split tables: the table rts_ctx_autotune keeps (if it keeps one), and a
    forced one (every tile that lived longer than 25 us in 4..8 pieces) under both packet kernels and as 8 interleaved stripes."""
    wl = workloads.prepare_config(name)
    W, H = wl.W, wl.H
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
        chosen, ms = ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        assert chosen in (3, 8)
        if name == "atrium_1080p":
            assert ctx.get_option("split_tiles") > 0                        # its frame is a few long waves: the table must win
        assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all(), "autotuned"
        for kernel in (3, 8):
            ctx.set_option("kernel", kernel)
            tiles, pieces = ctx.plan_splits(wl.constants, d_pos, W, H, d_mask, light=wl.light, min_life_us=25.0, piece_us=6.0, max_pieces=8)
            assert tiles > 0
            got = _device_frame(ctx, wl, d_pos, d_mask)
            assert (got == want).all(), (kernel, tiles, int((got != want).sum()))
        got = np.full((H, W), 9, np.uint8)
        ctx.h2d(d_mask, got)
        for r in range(8):
            ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light, stripes=(32, 8, r))
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, 8, r, light=wl.light)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        assert (got == want).all(), ("8 tuned stripes", int((got != want).sum()))
    finally:
        ctx.clear_splits()
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_fast_reciprocal_is_the_ieee_quotient_for_every_input_it_is_used_on(ctx):
    """The kernels compute 1.0f / x (comp:44 in the triangle test, comp:77 for 1/d) as v_rcp_f32 + one Newton step when a whole
    wave's values lie in [2^-100, 2^100), and by the general division otherwise.  The shortcut's exactness argument is this
    run: all 3 355 443 200 bit patterns of the range, on this device, against the division."""
    checked, differ, example = ctx.selftest_reciprocal()
    assert checked == 200 * (1 << 23) * 2
    assert differ == 0, f"{differ} inputs differ, e.g. 0x{example:08x}"


def test_non_finite_bvh_takes_the_exact_path(ctx):
    """A packed buffer from another producer may carry Inf boxes: the fast slab test must not be used."""
    sc = scenes.terrain(9)
    verts, idx = sc.flat()
    packed = api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes.copy()
    f = packed.view(np.float32)
    f[0, 0:3] = -np.inf                                                            # root box = everything
    f[1, 0:3] = np.inf
    rays = _random_rays(50000, 9, sc.bbox_min - 1, sc.bbox_max + 3)
    want, _, _ = oracle.trace_rays(packed, rays)
    ctx.set_bvh(packed)
    assert ctx.get_option("bvh_finite") == 0
    for v in _variants(ctx):
        ctx.set_option("kernel", v)
        assert (ctx.trace_rays(rays) == want).all()


def test_single_triangle_and_two_triangle_trees(ctx):
    for P in (1, 2):
        v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 2], [1, 0, 2], [0, 1, 2]], np.float32)[:3 * P]
        idx = np.arange(3 * P, dtype=np.uint32)
        packed = api.BVHBuilder().build(v, 3, idx, P).m_packedNodes
        rays = _random_rays(5000, P, np.float32(-1), np.float32(3))
        rays[:1000, 4:7] = [0, 0, -1]
        rays[:1000, 2] = 3
        want, _, _ = oracle.trace_rays(packed, rays)
        ctx.set_bvh(packed)
        for k in _variants(ctx):
            ctx.set_option("kernel", k)
            assert (ctx.trace_rays(rays) == want).all()
        assert want.min() == 0 and want.max() == 1


def test_median_split_tree_traces_identically(ctx):
    """A tree from the > 1 000 000-primitive branch of the builder (threshold lowered) through the kernels."""
    sc = scenes.terrain(40)
    verts, idx = sc.flat()
    packed = api.BVHBuilder(sah_prim_limit=100).build(verts, 8, idx, sc.triangle_count).m_packedNodes
    rays = _random_rays(100000, 4, sc.bbox_min - 1, sc.bbox_max + 4)
    want, _, _ = oracle.trace_rays(packed, rays)
    ctx.set_bvh(packed)
    for k in _variants(ctx):
        ctx.set_option("kernel", k)
        assert (ctx.trace_rays(rays) == want).all()


def test_error_behaviour(ctx):
    fresh = api.ShadowContext(0)
    try:
        k = api.RayTracingConstants.make([0, 0, 0], [0, 1, 0], 8, 8)
        with pytest.raises(api.RtsError) as e:
            fresh.trace_shadow_mask(k, np.zeros((8, 8, 4), np.float32), 8, 8)
        assert e.value.status == 4                                                 # RTS_ERR_NO_BVH
        with pytest.raises(api.RtsError) as e:
            fresh.set_bvh(np.zeros((7, 4), np.uint32))
        assert e.value.status == 5
        sc = scenes.terrain(5)
        verts, idx = sc.flat()
        fresh.set_bvh(api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes)
        with pytest.raises(api.RtsError) as e:
            fresh.trace_shadow_mask(k, np.zeros((8, 8, 4), np.float32), 8, 8, row_begin=5, row_end=9)
        assert e.value.status == 1
        with pytest.raises(api.RtsError):
            fresh.set_option("no_such_option", 1)
        d = fresh.malloc(64)
        try:                                      # frames beyond 2^31 pixels are refused before anything is touched
            with pytest.raises(api.RtsError) as e:
                fresh.trace_shadow_mask_device(k, d, 65536, 65536, d)
            assert e.value.status == 1
        finally:
            fresh.free(d)
        assert fresh.trace_rays(np.zeros((0, 8), np.float32)).size == 0            # empty input is fine
    finally:
        fresh.close()


def test_contexts_and_uploads_do_not_leak_device_memory(ctx):
    """Create/destroy contexts and re-upload streams of different sizes (host builder and device builder, host-pointer
    traces that grow the staging buffers): the device's free memory returns to where it was."""
    small, big = scenes.terrain(8), scenes.terrain(96)
    blobs = []
    for sc in (small, big):
        verts, idx = sc.flat()
        blobs.append((verts, idx, sc.triangle_count, api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes))
    k = api.RayTracingConstants.make([0, 50, 0], [0.3, 0.8, 0.5], 64, 64)
    pos = np.zeros((64, 64, 4), np.float32)

    def cycle():
        with api.ShadowContext(0) as c:
            for verts, idx, P, packed in blobs + blobs[::-1]:
                c.set_bvh(packed)
                c.trace_shadow_mask(k, pos, 64, 64)
                api.bvh_build_device(c, verts, 8, idx, P, install=True, want_packed=False)
                c.trace_shadow_mask(k, pos, 64, 64)
            c.set_option("wave_stats", 64)
            c.set_tile_order(np.arange(64, dtype=np.uint32))
    cycle()                                                  # first use may grow allocator pools / hipcub temporaries
    ctx.synchronize()
    before = ctx.mem_info()[0]
    for _ in range(5):
        cycle()
    ctx.synchronize()
    after = ctx.mem_info()[0]
    assert before - after < (8 << 20), f"device memory shrank by {(before - after) >> 20} MiB over 5 context cycles"


def test_tuning_options_never_change_the_mask(ctx):
    """Packet size, dissolve window/threshold (16 = always dissolve -> every ray finishes lane-per-ray from the
    node it stands or waits on; 0 = never), workgroup shape and block order are speed knobs only."""
    wl = workloads.prepare("atrium", 640, 360)
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                    oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
    ctx.set_bvh(wl.packed)
    defaults = {k: ctx.get_option(k) for k in ("packet_budget", "packet_share", "block_waves", "xcd_swizzle", "kernel", "row_order")}
    try:
        for kernel in (-1, 0, 3, 5, 7):                      # dispatch order of the tile rows: 2-D grids of every kernel family
            for order in (1, 2, 0):
                ctx.set_option("kernel", kernel)
                ctx.set_option("row_order", order)
                got = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
                assert (got == want).all(), (kernel, "row_order", order)
        for kernel in (-1, 3, 4, 5):
            for budget, share in ((1, 16), (1, 0), (2, 8), (8, 4), (64, 16), (64, 1)):
                for bw in (1, 4):
                    ctx.set_option("kernel", kernel)
                    ctx.set_option("packet_budget", budget)
                    ctx.set_option("packet_share", share)
                    ctx.set_option("block_waves", bw)
                    ctx.set_option("xcd_swizzle", (budget + bw) & 1)
                    got = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
                    assert (got == want).all(), (kernel, budget, share, bw)
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)


def test_auto_kernel_selection_and_names(ctx):
    small = workloads.prepare("cornell", 64, 64, via_obj=False)
    big = workloads.prepare("cornell", 640, 480, via_obj=False)
    ctx.set_option("kernel", -1)
    for wl, name in ((small, "shadowMaskKernel<7>"), (big, "shadowMaskPacketKernel<1>")):
        want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                        oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
        ctx.set_bvh(wl.packed)
        got = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
        assert (got == want).all()
        assert ctx.last_kernel_name() == name


def test_autotune_leaves_legal_options_and_the_same_mask(ctx):
    """rts_ctx_autotune picks a kernel, a dissolve threshold and a row order by timing them: whatever it picks, the mask is the
    oracle's, and the three options hold values it may pick."""
    try:
        for scene, W, H in (("atrium", 960, 540), ("cornell", 200, 120)):
            wl = workloads.prepare(scene, W, H, via_obj=False)
            want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                            oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
            ctx.set_bvh(wl.packed)
            ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
            d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
            ctx.h2d(d_pos, wl.positions)
            chosen, ms = ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            assert chosen in (3, 7, 8) and ctx.get_option("kernel") == chosen and ms > 0
            assert ctx.get_option("packet_share") in (4, 6) and ctx.get_option("row_order") in (0, 1)
            ctx.h2d(d_mask, np.full(W * H, 9, np.uint8))
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
            got = np.empty(W * H, np.uint8)
            ctx.d2h(got, d_mask)
            assert (got.reshape(H, W) == want).all()
            ctx.free(d_pos); ctx.free(d_mask)
    finally:
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)


def test_autotune_for_motion_keeps_only_the_block_sorted_table(ctx):
    """Option "tune_for_motion": the tuner may keep ONE kind of table -- the whole dispatch in table order, sorted by blocks of
    16 x 16 tiles, dealt over the XCDs by squares, no pieces -- or none.  The mask of the tuned frame, of the same table under a
    moved camera and of a moved light is the oracle's."""
    wl = workloads.prepare("atrium", 960, 540)
    W, H, sc = wl.W, wl.H, wl.scene
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)
    try:
        ctx.h2d(d_pos, wl.positions)
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
        ctx.set_option("tune_for_motion", 1)
        assert ctx.get_option("tune_for_motion") == 1
        ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        plan = ctx.split_plan()
        if plan is not None:
            assert plan["life_block"] == 16 and plan["xcd_square"] == 32 and plan["front_share"] == 1.0
            assert ctx.get_option("split_tiles") == 0 and ctx.get_option("front_tiles") == ((W + 7) // 8) * ((H + 7) // 8)
        want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
        assert (_device_frame(ctx, wl, d_pos, d_mask) == want).all()
        # the camera 2 % further along its view direction, the same table
        eye2 = (sc.eye + (sc.target - sc.eye) * np.float32(0.02)).astype(np.float32)
        pos2, _ = api.primary_positions(wl.packed, eye2, sc.target, sc.fovy, W, H)
        k2 = api.RayTracingConstants.make(eye2, sc.light_direction, W, H)
        want2, _, _ = oracle.shadow_mask(wl.packed, k2.as_array(), oracle.light_from_product(wl.light, k2), pos2, W, H)
        ctx.h2d(d_pos, pos2)
        got = np.full((H, W), 9, np.uint8)
        ctx.h2d(d_mask, got)
        ctx.trace_shadow_mask_device(k2, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        assert (got == want2).all(), int((got != want2).sum())
    finally:
        ctx.set_option("tune_for_motion", 0)
        ctx.clear_splits()
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
        ctx.free(d_pos); ctx.free(d_mask)


def test_planned_tile_order_never_changes_the_mask(ctx):
    """rts_ctx_plan_tile_order: soft shadows (no split table: four waves per tile) dispatched longest tile first, dealt over the
    XCDs by image squares.  The mask -- counts of unoccluded samples -- is the oracle's with the order on, off, sorted by blocks,
    for one stripe of three, and after the tuner has had its say."""
    wl = workloads.prepare("atrium", 640, 360)
    W, H = wl.W, wl.H
    light = api.Light.make(api.Light.POINT, wl.scene.light_point, scenes.jitter_offsets(16, 0.5, 3))
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(light, wl.constants), wl.positions, W, H)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    ctx.set_bvh(wl.packed)
    d_pos, d_mask = ctx.malloc(wl.positions.nbytes), ctx.malloc(W * H)

    def frame():
        got = np.full((H, W), 99, np.uint8)
        ctx.h2d(d_mask, got)
        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=light)
        ctx.synchronize()
        ctx.d2h(got, d_mask)
        return got

    try:
        ctx.h2d(d_pos, wl.positions)
        for kernel in (3, 8):
            ctx.set_option("kernel", kernel)
            for square, block in ((32, 0), (0, 0), (5, 4)):
                assert ctx.plan_tile_order(wl.constants, d_pos, W, H, d_mask, light=light, xcd_square=square, life_block=block) == tiles
                assert ctx.get_option("tile_order_tiles") == tiles and ctx.get_option("tile_order_square") == square
                assert (frame() == want).all(), (kernel, square, block)
            ctx.set_option("tile_order", 0)
            assert (frame() == want).all()
            ctx.set_option("tile_order", 1)
            # one stripe of three: an order for ITS dispatch; the full frame then runs without (another workgroup count)
            assert ctx.plan_tile_order(wl.constants, d_pos, W, H, d_mask, light=light, stripes=(32, 3, 1)) > 0
            got = np.full((H, W), 99, np.uint8)
            ctx.h2d(d_mask, got)
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, 3, 1, light=light)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            own = np.zeros(H, bool)
            for b, e in partition.stripe_rows(H, 3, 1, band=32, interleaved=True):
                own[b:e] = True
            assert (got[own] == want[own]).all() and (got[~own] == 99).all()
            assert (frame() == want).all()
        # an order planned on 16 samples leaves the one-sample frame of the same size its everyday launch
        ctx.set_option("kernel", 8)
        assert ctx.plan_tile_order(wl.constants, d_pos, W, H, d_mask, light=light) == tiles and ctx.get_option("tile_order_planned") == 1
        want1, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), oracle.light_from_product(wl.light, wl.constants), wl.positions, W, H)
        assert (_device_frame(ctx, wl, d_pos, d_mask) == want1).all() and ctx.last_kernel_name() == "shadowMaskPacketKernel<1,wide>"
        assert (frame() == want).all()
        # the tuner: plans an order for a dispatch of several samples, keeps it or not; a one-sample frame gets none
        ctx.set_option("kernel", -1)
        ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=light)
        assert ctx.get_option("tile_order_tiles") in (0, tiles)
        assert (frame() == want).all()
        ctx.autotune(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        assert ctx.get_option("tile_order_tiles") == 0
    finally:
        ctx.set_tile_order(None)
        ctx.clear_splits()
        ctx.set_option("kernel", -1); ctx.set_option("packet_share", 4); ctx.set_option("row_order", 0)
        ctx.free(d_pos); ctx.free(d_mask)


def test_mixed_sign_and_unordered_boxes_take_the_generic_slab_test(ctx):
    """Light inside the room: ray directions of one tile straddle the sign planes (generic form 8).  A blob whose
    inner boxes have bboxMin > bboxMax on an axis (another producer) must switch the ordered test off."""
    wl = workloads.prepare("cornell", 320, 320, via_obj=False)
    packed = wl.packed.copy()
    N = 2 * wl.prim_count - 1
    inner = np.nonzero(packed[0:2 * N:2, 3] == 0xFFFFFFFF)[0][5:40]
    for i in inner:                                                      # swap min.x <-> max.x on some inner nodes
        packed[2 * i, 0], packed[2 * i + 1, 0] = packed[2 * i + 1, 0], packed[2 * i, 0]
    for blob, ordered in ((wl.packed, 1), (packed, 0)):
        want, _, _ = oracle.shadow_mask(blob, wl.constants.as_array(),
                                        oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
        ctx.set_bvh(blob)
        assert ctx.get_option("bvh_ordered") == ordered
        for v in _variants(ctx):
            ctx.set_option("kernel", v)
            got = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
            assert (got == want).all(), (ordered, v)
    ctx.set_option("kernel", -1)


def test_wave_stats_diagnostics_do_not_touch_the_output(ctx):
    wl = workloads.prepare("cornell", 256, 256, via_obj=False)
    ctx.set_bvh(wl.packed)
    ctx.set_option("kernel", 3)
    a = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
    waves = (256 // 8) * (256 // 8)
    ctx.set_option("wave_stats", waves)
    b = ctx.trace_shadow_mask(wl.constants, wl.positions, wl.W, wl.H, light=wl.light)
    st = ctx.read_wave_stats(waves)
    ctx.set_option("wave_stats", 0)
    ctx.set_option("kernel", -1)
    assert (a == b).all()
    assert (st[:, 1] > st[:, 0]).all()                                   # every wave stamped start < end


def test_interleaved_stripes_single_dispatch(ctx):
    """configs[3], the form bench.py --scaling strong uses: each device traces its interleaved 32-row bands in
    ONE dispatch.  All stripes of a partition, traced one after the other into one mask, give the full frame;
    a single stripe touches only its own rows."""
    wl = workloads.prepare("atrium", 648, 500)                       # 500 rows: the last band is ragged
    want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                    oracle.light_from_product(wl.light, wl.constants), wl.positions, wl.W, wl.H)
    ctx.set_bvh(wl.packed)
    W, H = wl.W, wl.H
    d_pos = ctx.malloc(wl.positions.nbytes)
    d_mask = ctx.malloc(W * H)
    ctx.h2d(d_pos, wl.positions)
    try:
        for kernel in (-1, 0, 3, 5):
            ctx.set_option("kernel", kernel)
            for n in (1, 2, 3, 8):
                got = np.full((H, W), 7, np.uint8)
                ctx.h2d(d_mask, got)
                for stripe in range(n):
                    ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, n, stripe, light=wl.light)
                ctx.synchronize()
                ctx.d2h(got, d_mask)
                assert (got == want).all(), (kernel, n)
            got = np.full((H, W), 7, np.uint8)
            ctx.h2d(d_mask, got)
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, 4, 1, light=wl.light)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            own = np.zeros(H, bool)
            for b, e in partition.stripe_rows(H, 4, 1, band=32, interleaved=True):
                own[b:e] = True
            assert (got[own] == want[own]).all() and (got[~own] == 7).all()
        with pytest.raises(api.RtsError):
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 20, 2, 0)      # band not a multiple of 8
        with pytest.raises(api.RtsError):
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 32, 2, 2)      # stripe out of range
        ctx.set_option("kernel", 5)
        with pytest.raises(api.RtsError):
            ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 8, 2, 0)       # 16x16 tiles x 2x2 per workgroup need 32-row bands
        ctx.set_option("kernel", 3)                          # the default packet kernel (one 8x8 tile per workgroup) takes 8-row bands
        for n in (2, 5):
            got = np.full((H, W), 7, np.uint8)
            ctx.h2d(d_mask, got)
            for stripe in range(n):
                ctx.trace_shadow_mask_stripes_device(wl.constants, d_pos, W, H, d_mask, 8, n, stripe, light=wl.light)
            ctx.synchronize()
            ctx.d2h(got, d_mask)
            assert (got == want).all(), ("band 8", n)
    finally:
        ctx.set_option("kernel", -1)
        ctx.free(d_pos)
        ctx.free(d_mask)


@pytest.mark.parametrize("scene,W,H", [("atrium", 480, 270), ("cornell", 333, 201), ("city", 1920, 1080)])
def test_gbuffer_pass_on_the_gpu_is_bit_identical_to_the_oracle(ctx, scene, W, H):
    """the G-buffer generator on the GPU (producer of binding 2; contract Model.frag:35-39) against the
    INDEPENDENT closest-hit oracle (oracle/rts_oracle.cpp: orc_primary_gbuffer) and against the host pass: camera-relative
    positions and normals equal bit for bit, every texel."""
    wl = workloads.prepare(scene, W, H, via_obj=False)
    sc = wl.scene
    want_pos, want_nrm, want_hits = oracle.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    pos, nrm, hits = api.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    assert hits == want_hits
    assert (pos.view(np.uint32) == want_pos.view(np.uint32)).all() and (nrm.view(np.uint32) == want_nrm.view(np.uint32)).all()
    ctx.set_bvh(wl.packed)
    d_pos = ctx.malloc(pos.nbytes)
    d_nrm = ctx.malloc(nrm.nbytes)
    try:
        api.primary_gbuffer_device(ctx, sc.eye, sc.target, sc.fovy, W, H, d_pos, d_nrm)
        ctx.synchronize()
        gpos, gnrm = np.zeros_like(pos), np.zeros_like(nrm)
        ctx.d2h(gpos, d_pos)
        ctx.d2h(gnrm, d_nrm)
        bad = (gpos.view(np.uint32) != want_pos.view(np.uint32)).any(axis=2)
        assert not bad.any(), f"{bad.sum()} of {bad.size} position texels differ from the oracle, first at {np.argwhere(bad)[:3].tolist()}"
        badn = (gnrm.view(np.uint32) != want_nrm.view(np.uint32)).any(axis=2)
        assert not badn.any(), f"{badn.sum()} of {badn.size} normal texels differ from the oracle"
        # and the shadow mask traced from the device-made G-buffer equals the oracle's on the same buffer
        d_mask = ctx.malloc(W * H)
        ctx.set_option("kernel", -1)
        ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=wl.light)
        ctx.synchronize()
        got = np.zeros((H, W), np.uint8)
        ctx.d2h(got, d_mask)
        ctx.free(d_mask)
        want, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(),
                                        oracle.light_from_product(wl.light, wl.constants), gpos, W, H)
        assert (got == want).all()
    finally:
        ctx.free(d_pos)
        ctx.free(d_nrm)


def test_whole_frame_on_the_device_gbuffer_mask_combine_and_blob(ctx, tmp_path):
    """SURVEY.md 8 f4 on the GPU: G-buffer -> shadow mask -> combine pass (Combine.frag:18-37) without leaving the device,
    against the oracle's combine (orc_combine) of the oracle's mask; the packed stream goes through the blob file first.
    Point light, 16-sample soft shadows and the reference's directional light."""
    wl = workloads.prepare("atrium", 640, 360)
    W, H, sc = wl.W, wl.H, wl.scene
    blob = api.load_bvh(api.save_bvh(str(tmp_path / "atrium.bvh"), wl.packed))
    assert (blob == wl.packed).all()
    ctx.set_bvh(blob)
    ctx.set_option("kernel", -1)
    pos, nrm, _ = oracle.primary_gbuffer(wl.packed, sc.eye, sc.target, sc.fovy, W, H)
    d_pos, d_nrm, d_mask, d_rgb = ctx.malloc(pos.nbytes), ctx.malloc(nrm.nbytes), ctx.malloc(W * H), ctx.malloc(W * H * 3)
    try:
        api.primary_gbuffer_device(ctx, sc.eye, sc.target, sc.fovy, W, H, d_pos, d_nrm)
        for light in (wl.light, workloads.relight(wl, "point", 16).light, None):
            ctx.trace_shadow_mask_device(wl.constants, d_pos, W, H, d_mask, light=light)
            api.combine_device(ctx, wl.constants, light, d_pos, d_nrm, d_mask, W, H, d_rgb)
            ctx.synchronize()
            got = np.zeros((H, W, 3), np.uint8)
            ctx.d2h(got, d_rgb)
            olight = oracle.light_from_product(light, wl.constants)
            mask, _, _ = oracle.shadow_mask(wl.packed, wl.constants.as_array(), olight, pos, W, H)
            # the checker is the ORACLE's restatement of Combine.frag:18-37 (orc_combine), not the product's own host build
            want = oracle.combine(wl.constants.as_array(), olight if light is not None else None, pos, nrm, mask)
            assert (got == want).all(), f"{(got != want).any(axis=2).sum()} pixels differ"
            assert want.max() > 100 and (want[nrm[..., :3].any(axis=2)] >= int(0.15 * 255)).all()
        with pytest.raises(api.RtsError):
            api.combine_device(ctx, wl.constants, wl.light, None, d_nrm, d_mask, W, H, d_rgb)   # point light needs positions
    finally:
        for d in (d_pos, d_nrm, d_mask, d_rgb):
            ctx.free(d)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_scenes_cameras_and_options(ctx, seed):
    """Property test: whatever the scene, camera, light, frame size, sample count and tuning knobs, every kernel's
    mask equals the oracle's.  Exercises mixed sign patterns, sparse tiles, early-dying packets and dissolving."""
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([1, 2, 5, 40, 300, 3000]))
    kind = seed % 3
    if kind == 0:                                                     # random soup
        c = rs.random_sample((n, 1, 3)) * 20 - 10
        verts = (c + (rs.random_sample((n, 3, 3)) - 0.5) * rs.choice([0.2, 2.0, 8.0])).astype(np.float32).reshape(-1, 3)
    elif kind == 1:                                                   # many coplanar, axis-aligned, touching triangles
        g = rs.randint(0, 6, size=(n, 1, 3)).astype(np.float32)
        verts = (g + rs.randint(0, 2, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    else:                                                             # huge coordinates + tiny triangles
        c = (rs.random_sample((n, 1, 3)) - 0.5) * 2e4
        verts = (c + (rs.random_sample((n, 3, 3)) - 0.5) * 1e-2).astype(np.float32).reshape(-1, 3)
    idx = np.arange(verts.shape[0], dtype=np.uint32)
    packed = api.BVHBuilder().build(verts, 3, idx, n).m_packedNodes
    lo, hi = verts.min(0), verts.max(0)
    W, H = int(rs.randint(1, 300)), int(rs.randint(1, 200))
    eye = (hi + (hi - lo) * rs.random_sample(3) + 1).astype(np.float32)
    target = (lo + (hi - lo) * rs.random_sample(3)).astype(np.float32)
    pos, _ = api.primary_positions(packed, eye, target, 1.0, W, H)
    k = api.RayTracingConstants.make(eye, [0.3, 0.8, 0.5], W, H)
    lights = [None,
              api.Light.make(api.Light.POINT, (lo + (hi - lo) * rs.random_sample(3)).astype(np.float32)),   # inside the scene
              api.Light.make(api.Light.DIRECTIONAL, [0, 1, 0]),                                                # axis-parallel: EXACT path
              api.Light.make(api.Light.POINT, hi + 5, scenes.jitter_offsets(int(rs.randint(2, 9)), 0.7, seed)),
              api.Light.make(api.Light.POINT, hi + 5, scenes.jitter_offsets(int(rs.randint(9, 65)), 0.7, seed),
                             nsamples=int(rs.randint(2, 9)))]                                                  # per-pixel jitter
    ctx.set_bvh(packed)
    d_pos, d_mask = ctx.malloc(pos.nbytes), ctx.malloc(W * H)
    try:
        for light in lights:
            want, _, _ = oracle.shadow_mask(packed, k.as_array(), oracle.light_from_product(light, k), pos, W, H)
            for kernel in range(ctx.get_option("kernel_count")):
                ctx.set_option("kernel", kernel)
                ctx.set_option("packet_budget", int(rs.choice([1, 2, 8, 50])))
                ctx.set_option("packet_share", int(rs.choice([0, 2, 4, 9, 16])))
                ctx.set_option("block_waves", int(rs.choice([1, 4])))
                ctx.set_option("wide_lane", int(rs.randint(0, 2)))
                ctx.set_option("soft_split", int(rs.randint(0, 2)))
                got = ctx.trace_shadow_mask(k, pos, W, H, light=light)
                bad = int((got != want).sum())
                assert bad == 0, (seed, n, W, H, kernel, "light", lights.index(light), bad)
                # ... through a split table with random thresholds (one-sample lights, the two packet kernels, one-tile workgroups)
                if kernel in (3, 8) and (light is None or light.nsamples <= 1) and ctx.get_option("wide_nodes") > 0:
                    ctx.set_option("block_waves", 1)
                    ctx.set_option("wide_lane", 0)
                    ctx.h2d(d_pos, pos)
                    tiles, _ = ctx.plan_splits(k, d_pos, W, H, d_mask, light=light, min_life_us=float(rs.choice([0.3, 1.0, 3.0])),
                                               piece_us=float(rs.choice([0.2, 1.0])), max_pieces=int(rs.choice([2, 5, 16])),
                                               front_share=float(rs.choice([0.0, 0.4, 1.0])), xcd_square=int(rs.choice([0, 2, 32])),
                                               life_block=int(rs.choice([0, 3, 16])))
                    got = np.full((H, W), 9, np.uint8)
                    ctx.h2d(d_mask, got)
                    ctx.trace_shadow_mask_device(k, d_pos, W, H, d_mask, light=light)
                    ctx.synchronize()
                    ctx.d2h(got, d_mask)
                    ctx.clear_splits()
                    assert (got == want).all(), (seed, kernel, "split table", tiles, int((got != want).sum()))
                # ... and as row stripes (contiguous through the host-pointer entry; 32-row interleaved bands in one dispatch)
                ns = int(rs.randint(2, 5))
                got = np.full((H, W), 9, np.uint8)
                for r in range(ns):
                    for b, e in partition.stripe_rows(H, ns, r, 32, False):
                        ctx.trace_shadow_mask(k, pos, W, H, light=light, row_begin=b, row_end=e, out=got)
                assert (got == want).all(), (seed, kernel, "contiguous stripes", ns)
                got = np.full((H, W), 9, np.uint8)
                ctx.h2d(d_pos, pos)
                ctx.h2d(d_mask, got)
                for r in range(ns):
                    ctx.trace_shadow_mask_stripes_device(k, d_pos, W, H, d_mask, 32, ns, r, light=light)
                ctx.synchronize()
                ctx.d2h(got, d_mask)
                assert (got == want).all(), (seed, kernel, "interleaved stripes", ns)
    finally:
        for key, v in (("kernel", -1), ("packet_budget", 16), ("packet_share", 4), ("block_waves", 1), ("wide_lane", 0), ("soft_split", 1)):
            ctx.set_option(key, v)
        ctx.free(d_pos)
        ctx.free(d_mask)


def test_plain_c_caller_of_the_consumer_seam(tmp_path):
    """A C99 program (tests/c/seam2_gpu.c) that does what a replacement of renderShadowMaskCompute does
    (RayTracedShadows.cpp:570-595): rts_ctx_create -> rts_ctx_set_bvh -> device buffers -> rts_trace_shadow_mask_device ->
    read back, for every kernel variant, against the committed golden masks of tests/golden/cornell_128.npz (the
    reference's directional light and the point light).  Compiled here with the box's C compiler."""
    import shutil
    import struct
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler on this box")
    g = np.load(os.path.join(GOLD, "cornell_128.npz"))
    H, W = g["mask_dir"].shape
    packed = np.ascontiguousarray(g["packed"], np.uint32).reshape(-1, 4)
    blob = tmp_path / "cornell_128.bin"
    with open(blob, "wb") as fh:
        fh.write(struct.pack("<3I", W, H, packed.shape[0]))
        fh.write(packed.tobytes())
        fh.write(np.ascontiguousarray(g["constants"], np.float32).tobytes())
        fh.write(np.ascontiguousarray(g["positions"], np.float32).tobytes())
        fh.write(np.ascontiguousarray(g["light_point"], np.float32).tobytes())
        fh.write(np.ascontiguousarray(g["mask_dir"], np.uint8).tobytes())
        fh.write(np.ascontiguousarray(g["mask_point"], np.uint8).tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "seam2_gpu")
    libdir = os.path.dirname(api.lib_path())
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "seam2_gpu.c"), "-o", exe, "-L", libdir, "-lrts", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe, str(blob)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 2 * (1 + api.ShadowContext(0).get_option("kernel_count"))
