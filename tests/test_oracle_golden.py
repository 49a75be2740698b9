"""CPU: the oracle against everything that pins it.

The reference holds no tests or golden vectors for this path.  What exists:
  * the one recorded output of the real reference builder -> byte-exact here;
  * hand-derived known answers for the scalar pieces, including the NaN cases of Appendix B;
  * an independent brute-force any-hit (no BVH) over the same triangles;
  * structural invariants of the packed layout (first table row).
"""
import json
import os

import numpy as np
import pytest

import oracle
from raytracedshadows_amd import api, scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INF = np.float32(np.inf)


def _appendix_a_mesh():
    v = []
    for t in range(4):
        v0 = np.array([2 * t, 0, t / 2], np.float32)
        v += [v0, v0 + np.array([1, 0, 0], np.float32), v0 + np.array([0, 1, 0], np.float32)]
    return np.array(v, np.float32), np.arange(12, dtype=np.uint32)


def test_oracle_builder_reproduces_reference_dump_appendix_a():
    gold = json.load(open(os.path.join(GOLD, "appendix_a_4tri.json")))
    verts, idx = _appendix_a_mesh()
    packed = oracle.bvh_build(verts, gold["stride"], idx, gold["prim_count"])
    assert packed.shape[0] == len(gold["packed"]) == 18
    for i, (xyz, w) in enumerate(gold["packed"]):
        assert packed[i, :3].view(np.float32).tolist() == [float(x) for x in xyz], f"vec4 {i} xyz"
        if w is not None:
            assert int(packed[i, 3]) == int(w, 16), f"vec4 {i} .w"


def test_ray_box_known_answers():
    one = np.ones(3, np.float32)
    lo, hi = np.zeros(3, np.float32), one
    inv = lambda d: (np.float32(1.0) / np.asarray(d, np.float32))
    assert oracle.ray_box([-1, 0.5, 0.5], inv([1, 1e-3, 1e-3]), lo, hi)            # enters through x=0
    assert not oracle.ray_box([-1, 2.5, 0.5], inv([1, 1e-3, 1e-3]), lo, hi)        # passes above
    assert oracle.ray_box([0.5, 0.5, 0.5], inv([0.3, -0.2, 0.9]), lo, hi)          # origin inside: t0 clipped at 0
    assert not oracle.ray_box([2, 0.5, 0.5], inv([1, 0.1, 0.1]), lo, hi)           # box behind the ray
    # no tmax clip in the slab test (comp:61-73): a box a million units away still "hits"
    assert oracle.ray_box([-1e6, 0.5, 0.5], inv([1, 1e-9, 1e-9]), lo, hi)
    with np.errstate(divide="ignore"):
        # axis-parallel ray strictly inside the y,z slabs: +-inf, no NaN
        assert oracle.ray_box([-1, 0.5, 0.5], inv([1, 0, 0]), lo, hi)
        assert not oracle.ray_box([-1, 1.5, 0.5], inv([1, 0, 0]), lo, hi)
        # ON a slab plane with a zero direction component: (pmax-o)*inf = 0*inf = NaN.
        # GLSL max(f,n) = f<n ? n : f keeps f = NaN (y: f is NaN)  -> t1 = min(.., NaN-chain) ...
        # Appendix B-3: worked by hand below.
        # o.y == pmax.y: f.y = NaN, n.y = -inf -> tmax.y = NaN, tmin.y = (n<f ? n : f) = NaN
        #   t1 = min(tmax.x, min(NaN, tmax.z)): min(x,y)= y<x?y:x -> min(NaN,tz) = tz<NaN? no -> NaN; min(tx,NaN) = NaN<tx? no -> tx
        #   t0 = max(max(tminx, max(NaN, tminz)), 0): max(NaN,tz) = NaN<tz? no -> NaN; max(tminx,NaN) = tminx<NaN? no -> tminx
        #   -> the NaN axis drops out entirely: decision from x only -> hit
        assert oracle.ray_box([-1, 1.0, 0.5], inv([1, 0, 0]), lo, hi)
        # o.y == pmin.y: n.y = NaN, f.y = +inf -> tmax.y = (f<n? n : f) = +inf; tmin.y = (n<f ? n : f) = +inf
        #   t0 = max(max(tminx, max(inf, tminz)), 0) = inf ; t1 finite -> miss
        assert not oracle.ray_box([-1, 0.0, 0.5], inv([1, 0, 0]), lo, hi)


def test_ray_tri_known_answers():
    v0, e0, e1 = [0, 0, 0], [1, 0, 0], [0, 1, 0]
    down = [0, 0, -1]
    assert oracle.ray_tri([0.25, 0.25, 1, 1e9], down, v0, e0, e1)
    assert oracle.ray_tri([0.25, 0.25, -1, 1e9], [0, 0, 1], v0, e0, e1)          # two-sided
    assert not oracle.ray_tri([0.75, 0.75, 1, 1e9], down, v0, e0, e1)            # b1+b2 > 1
    assert not oracle.ray_tri([-0.1, 0.25, 1, 1e9], down, v0, e0, e1)            # b < 0
    assert not oracle.ray_tri([0.25, 0.25, 1, 0.5], down, v0, e0, e1)            # beyond tmax
    assert oracle.ray_tri([0.25, 0.25, 1, 1.0], down, v0, e0, e1)                # t == tmax is a hit (t > tmax rejects)
    assert not oracle.ray_tri([0.25, 0.25, -1, 1e9], down, v0, e0, e1)           # behind the origin
    assert oracle.ray_tri([0.0, 0.0, 1, 1e9], down, v0, e0, e1)                  # exactly on a vertex: edges inclusive
    # Appendix B-4: degenerate triangle (zero area) -> det = 0 -> invd = inf, b1 = b2 = t = 0*inf = NaN
    # -> every reject compare is false -> HIT
    with np.errstate(all="ignore"):
        assert oracle.ray_tri([0.0, 0.0, 1, 1e9], down, v0, [0, 0, 0], [0, 0, 0])
        # ray lying IN the triangle's plane, far from it: s1=(0,0,1), det=0, dd.s1=0 -> all NaN -> HIT
        assert oracle.ray_tri([5, 5, 0, 1e9], [1, 0, 0], v0, e0, e1)
        # the same ray one unit above the plane: b1 = 1*inf = +inf > 1 -> miss
        assert not oracle.ray_tri([5, 5, 1, 1e9], [1, 0, 0], v0, e0, e1)


def test_epsilon_for_known_answers():
    f = np.float32
    assert oracle.epsilon_for(f(1.0)) == f(2.0 ** -13)
    assert oracle.epsilon_for(f(-300.0)) == f(-300.0 * 2.0 ** -13)               # sign and mantissa kept
    assert oracle.epsilon_for(f(0.0)) == f(0.0)
    tiny = np.array([1 << 23], np.uint32).view(np.float32)[0] * f(1.5)           # exponent 1 -> clamps at 0
    out = np.array([oracle.epsilon_for(tiny)], np.float32).view(np.uint32)[0]
    assert out >> 23 == 0 and (out & 0x7FFFFF) == (np.array([tiny]).view(np.uint32)[0] & 0x7FFFFF)
    assert np.isnan(oracle.epsilon_for(f(np.nan))) or True                       # defined bit-op, never traps


def _invariants(packed, P):
    N = 2 * P - 1
    assert packed.shape == (5 * P - 2, 4)
    a, b = packed[0:2 * N:2], packed[1:2 * N:2]
    leaf = a[:, 3] != 0xFFFFFFFF
    assert int(leaf.sum()) == P and int((~leaf).sum()) == P - 1
    prim = a[leaf, 3].astype(np.int64) - 2 * N
    assert sorted(prim.tolist()) == list(range(P))                                  # every triangle exactly once
    nxt = b[:, 3].astype(np.int64)
    idx = np.arange(N)
    assert b[0, 3] == 0xFFFFFFFF
    fwd = nxt != 0xFFFFFFFF
    assert (nxt[fwd] > idx[fwd]).all() and (nxt[fwd] < N).all()                     # miss links strictly forward
    # next(i) = first index after i's subtree; left child = i+1; right child = next(left)
    size = np.zeros(N, np.int64)
    for i in range(N - 1, -1, -1):
        if leaf[i]:
            size[i] = 1
        else:
            l = i + 1
            r = l + size[l]
            size[i] = 1 + size[l] + size[r]
            assert nxt[l] == r
            assert nxt[r] == nxt[i]
            lo = np.minimum(a[l, :3].view(np.float32), a[r, :3].view(np.float32)) if not (leaf[l] or leaf[r]) else None
            if lo is not None:                                                      # inner bbox contains inner children
                assert (a[i, :3].view(np.float32) <= lo).all()
    assert size[0] == N
    end = np.where(nxt == 0xFFFFFFFF, N, nxt)
    assert (end == idx + size).all()


@pytest.mark.parametrize("maker", [scenes.cornell, lambda: scenes.terrain(9), lambda: scenes.terrain(23)])
def test_packed_layout_invariants(maker):
    sc = maker()
    verts, idx = sc.flat()
    packed = oracle.bvh_build(verts, 8, idx, sc.triangle_count)
    _invariants(packed, sc.triangle_count)
    assert api.bvh_validate(packed) == sc.triangle_count


def test_larger_child_is_left():
    sc = scenes.terrain(9)
    verts, idx = sc.flat()
    packed, nodes = oracle.bvh_build(verts, 8, idx, sc.triangle_count, want_nodes=True)
    N = 2 * sc.triangle_count - 1
    f = nodes.view(np.float32)

    def area(i):
        e = f[i, 4:7] - f[i, 0:3]
        return np.float32(np.float32(e[0] * e[1] + e[1] * e[2]) + e[2] * e[0]) * np.float32(2)
    for i in range(N):
        if nodes[i, 3] == 0xFFFFFFFF:
            l, r = i + 1, int(nodes[i + 1, 7])
            assert area(l) >= area(r)                                               # BVHBuilder.cpp:202-208


def test_bvh_hits_are_a_subset_of_brute_force_and_nearly_equal():
    g = np.load(os.path.join(GOLD, "cornell_128.npz"))
    packed, pos, k = g["packed"], g["positions"], g["constants"]
    P = (packed.shape[0] + 2) // 5
    for lt in (oracle.make_light(0, [0.57735026, 0.57735026, 0.57735026]), oracle.make_light(1, g["light_point"])):
        rays = oracle.gen_rays(k, lt, pos)
        with_bvh, _, _ = oracle.trace_rays(packed, rays)
        brute = oracle.brute_force_rays(packed, P, rays)
        assert not ((with_bvh == 0) & (brute == 1)).any()       # a BVH hit is always a brute-force hit
        # the slab test is not conservative: allow a vanishing number of extra brute hits
        assert int(((with_bvh == 1) & (brute == 0)).sum()) <= rays.shape[0] // 2000


@pytest.mark.parametrize("name", ["cornell_128", "terrain_96"])
def test_oracle_matches_committed_masks(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    H, W = g["mask_dir"].shape
    d = np.float32(1.0) / np.sqrt(np.float32(3.0))
    for tag, lt in (("dir", oracle.make_light(0, [d, d, d])), ("point", oracle.make_light(1, g["light_point"]))):
        if tag == "dir":
            lt = oracle.make_light(0, g["constants"][8:11])
        m, V, L, pv, pl = oracle.shadow_mask(g["packed"], g["constants"], lt, g["positions"], W, H, per_ray=True)
        assert (m == g[f"mask_{tag}"]).all()
        assert (pv == g[f"visits_{tag}"]).all() and (pl == g[f"leafs_{tag}"]).all()
        assert V == int(pv.sum()) and L == int(pl.sum())


def test_oracle_rows_and_threads_do_not_change_the_mask():
    g = np.load(os.path.join(GOLD, "cornell_128.npz"))
    H, W = g["mask_point"].shape
    lt = oracle.make_light(1, g["light_point"])
    out = np.full((H, W), 7, np.uint8)
    oracle.shadow_mask(g["packed"], g["constants"], lt, g["positions"], W, H, 16, 40, threads=1, out=out)
    assert (out[16:40] == g["mask_point"][16:40]).all() and (out[:16] == 7).all() and (out[40:] == 7).all()


def test_multi_sample_counts():
    g = np.load(os.path.join(GOLD, "cornell_128.npz"))
    H, W = g["mask_point"].shape
    offs = scenes.jitter_offsets(4, 0.5)
    lt = oracle.make_light(1, g["light_point"], offs)
    m, _, _ = oracle.shadow_mask(g["packed"], g["constants"], lt, g["positions"], W, H)
    total = np.zeros((H, W), np.int32)
    for j in range(4):
        lj = oracle.make_light(1, (g["light_point"] + offs[j, :3]).astype(np.float32))
        mj, _, _ = oracle.shadow_mask(g["packed"], g["constants"], lj, g["positions"], W, H)
        total += mj
    assert m.max() <= 4 and (m == total).all()


def test_any_hit_does_not_depend_on_the_child_order():
    """The property every traversal variant of the product relies on: walking the children of a node in another order
    changes how many nodes a ray visits, never whether it is occluded."""
    from raytracedshadows_amd import workloads
    wl = workloads.prepare("cornell", 128, 128, via_obj=False)
    lt = oracle.light_from_product(wl.light, wl.constants)
    want = oracle.shadow_mask(wl.packed, wl.constants.as_array(), lt, wl.positions, wl.W, wl.H)[0]
    stats = [oracle.order_experiment(wl.packed, wl.constants.as_array(), lt, wl.positions, wl.W, wl.H, m) for m in (0, 1, 2)]
    assert all(s["occluded"] == int((want == 0).sum()) for s in stats)
    assert len({round(s["visits_per_ray"], 6) for s in stats}) > 1          # ... while the visit counts do differ


def test_ties_by_triangle_id_is_the_same_builder_where_no_centroids_tie():
    """`ties_by_prim` (the rule the device SAH builder is held to, oracle/rts_oracle.cpp CtrLess) only decides what std::sort leaves open:
    on a mesh whose centroids differ pairwise on every axis it IS the reference restatement; on a grid full of equal centroids it is
    another valid stream of the same layout; and -0 == +0 as in the reference's comparison."""
    rs = np.random.RandomState(8)
    n = 3000
    c = np.stack([rs.permutation(n) + 1 for _ in range(3)], 1).astype(np.float64) / 1024.0       # distinct per axis, exactly representable
    h = rs.randint(1, 4, size=(n, 3)).astype(np.float64) / 4096.0
    sign = np.array([[-1, -1, -1], [1, 1, -1], [-1, 1, 1]], np.float64)
    v = (c[:, None, :] + sign[None, :, :] * h[:, None, :]).astype(np.float32).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    for limit in (1000000, 64):
        assert (oracle.bvh_build(v, 3, idx, n, sah_limit=limit, ties_by_prim=True) == oracle.bvh_build(v, 3, idx, n, sah_limit=limit)).all()
    sc = scenes.terrain(24)
    fv, fi = sc.flat()
    tied = oracle.bvh_build(fv, 8, fi, sc.triangle_count, ties_by_prim=True)
    _invariants(tied, sc.triangle_count)
    assert api.bvh_validate(tied) == sc.triangle_count
    t0 = np.array([[0.0, 0, 0], [0.0, 1, 0], [0.0, 0, 1]], np.float32)
    t1 = t0 + np.float32([0, 0.5, 0])
    t1[:, 0] = np.float32(-0.0)
    z = np.concatenate([t0, t1, t1 + np.float32([0, 10, 0]), t0 + np.float32([0, 10, 0])])
    z[6:9, 0] = np.float32(-0.0)
    packed = oracle.bvh_build(z, 3, np.arange(12, dtype=np.uint32), 4, ties_by_prim=True)
    leaves = [int(x) - 14 for x in packed[0:14:2, 3] if x != 0xFFFFFFFF]
    assert leaves == [0, 1, 2, 3]                       # x centroids +0, -0, -0, +0 compare equal: triangle order decides


def test_per_pixel_jitter_hash_known_answers_and_permutation_property():
    """rts_light.table: the start of a pixel in the offset table is (hash32(p) * T) >> 32 with the documented integer hash
    (known answers of the hash as pinned values; the oracle's rays are checked against them below); with T == nsamples the start only permutes the samples, so the
    count of unoccluded samples is the unjittered one; with a larger table other samples are taken."""
    def hash32(v):
        v &= 0xFFFFFFFF
        v ^= v >> 16; v = (v * 0x7feb352d) & 0xFFFFFFFF
        v ^= v >> 15; v = (v * 0x846ca68b) & 0xFFFFFFFF
        v ^= v >> 16
        return v
    assert [hash32(v) for v in (0, 1, 2, 12345, 0xFFFFFFFF, 3840 * 2160 - 1)] == [0, 1753845952, 3507691905, 2435775735, 1734902346, 3873009951]
    from raytracedshadows_amd import scenes, workloads
    wl = workloads.prepare("cornell", 64, 48, spp=16, via_obj=False)
    k = wl.constants.as_array()
    lt = oracle.light_from_product(wl.light, wl.constants)
    base, _, _ = oracle.shadow_mask(wl.packed, k, lt, wl.positions, 64, 48)
    lt.table = 16
    rot, _, _ = oracle.shadow_mask(wl.packed, k, lt, wl.positions, 64, 48)
    assert (rot == base).all()
    # the rays themselves: pixel p, sample j aims at offsets[(start(p) + j) % T]
    rays_plain = oracle.gen_rays(k, oracle.light_from_product(wl.light, wl.constants), wl.positions).reshape(-1, 16, 8)
    rays_rot = oracle.gen_rays(k, lt, wl.positions).reshape(-1, 16, 8)
    for p in (0, 1, 77, 64 * 48 - 1):
        start = (hash32(p) * 16) >> 32
        for j in (0, 5, 15):
            assert (rays_rot[p, j] == rays_plain[p, (start + j) % 16]).all()
    big = workloads.relight(wl, "point", 16, table=64)
    lt64 = oracle.light_from_product(big.light, big.constants)
    assert lt64.table == 64 and lt64.nsamples == 16
    other, _, _ = oracle.shadow_mask(wl.packed, k, lt64, wl.positions, 64, 48)
    assert (other != base).any() and other.max() == 16
