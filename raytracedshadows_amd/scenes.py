"""Harness: seeded procedural stand-ins for the scenes BASELINE.json names.

No mesh assets exist offline, so each config gets a generator that emits an indexed
triangle mesh which is then written as real ``.obj`` text and read back through the OBJ reader
(``rtsh_obj_load``), i.e. the same ingest route the reference uses
(``Source/RayTracedShadows.cpp:764-824``).

=========  =====================================  ==========  =========================
name       stand-in for                            triangles   BASELINE.json config
=========  =====================================  ==========  =========================
cornell    Cornell-box-scale OBJ                   1 024       configs[0]  256 x 256
atrium     Sponza-class OBJ                        ~250 k      configs[1]  1920 x 1080
city       ~1M-tri OBJ, below the builder switch   999 488     configs[2-4]  3840 x 2160
city_big   same, above 1 000 000 primitives        1 034 288   builder's median-split branch
courtyard  San-Miguel-class OBJ (arcades, trees     999 990     configs[2-4] on the hard case: 60 nodes/ray,
           of leaf cards, furniture)                            tiles whose rays scatter between leaves
=========  =====================================  ==========  =========================

Camera and light follow the reference's defaults where it has any (eye = bbox.max + 2 looking at the
bbox centre, fov 1.0 rad: ``Source/RayTracedShadows.cpp:238-242``; directional light
normalize(1,1,1): ``cpp:245``); the in-scene point light is this harness's choice.
"""
import os

import numpy as np


# ------------------------------------------------------------------------------------------------
# mesh pieces (float64 maths, cast to float32 once at the end)
# ------------------------------------------------------------------------------------------------
def _grid(origin, du, dv, nu, nv):
    """(nu x nv) quads spanning origin + s*du + t*dv, s,t in [0,1]; two triangles per quad."""
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="xy")
    verts = (np.asarray(origin)[None, None, :] + S[..., None] * np.asarray(du)[None, None, :]
             + T[..., None] * np.asarray(dv)[None, None, :]).reshape(-1, 3)
    i = np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]
    i = i.reshape(-1)
    faces = np.stack([np.stack([i, i + 1, i + nu + 2], 1), np.stack([i, i + nu + 2, i + nu + 1], 1)], 1).reshape(-1, 3)
    return verts, faces


def _box(lo, hi, sub=1):
    """Axis-aligned box, each face a sub x sub grid."""
    lo = np.asarray(lo, float)
    hi = np.asarray(hi, float)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    parts = [(lo, ex, ey), (lo + ez, ey, ex), (lo, ey, ez), (lo + ex, ez, ey), (lo, ez, ex), (lo + ey, ex, ez)]
    return _merge([_grid(o, a, b, sub, sub) for (o, a, b) in parts])


def _boxes_fast(lo, hi):
    """Many 12-triangle boxes at once: lo/hi are (n,3)."""
    n = lo.shape[0]
    corners = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
    verts = (lo[:, None, :] + corners[None, :, :] * (hi - lo)[:, None, :]).reshape(-1, 3)
    quad = np.array([[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [2, 3, 7, 6], [1, 2, 6, 5], [0, 4, 7, 3]])
    tri = np.concatenate([quad[:, [0, 1, 2]], quad[:, [0, 2, 3]]], 0)
    faces = (tri[None, :, :] + 8 * np.arange(n)[:, None, None]).reshape(-1, 3)
    return verts, faces


def _cylinder(center, radius, y0, y1, sides, segments):
    a = np.linspace(0.0, 2.0 * np.pi, sides, endpoint=False)
    ys = np.linspace(y0, y1, segments + 1)
    ring = np.stack([center[0] + radius * np.cos(a), np.zeros_like(a), center[1] + radius * np.sin(a)], 1)
    verts = np.repeat(ring[None, :, :], segments + 1, 0)
    verts[:, :, 1] = ys[:, None]
    verts = verts.reshape(-1, 3)
    i = (np.arange(sides)[None, :] + sides * np.arange(segments)[:, None]).reshape(-1)
    j = ((np.arange(sides) + 1) % sides)[None, :] + sides * np.arange(segments)[:, None]
    j = j.reshape(-1)
    faces = np.stack([np.stack([i, j, j + sides], 1), np.stack([i, j + sides, i + sides], 1)], 1).reshape(-1, 3)
    return verts, faces


def _merge(parts):
    verts, faces, base = [], [], 0
    for v, f in parts:
        verts.append(v)
        faces.append(f + base)
        base += v.shape[0]
    return np.concatenate(verts, 0), np.concatenate(faces, 0)


def _value_noise(x, z, seed, cells):
    """Smooth lattice noise in [0,1] from a seeded table (bilinear + smoothstep)."""
    rs = np.random.RandomState(seed)
    table = rs.random_sample((cells + 2, cells + 2))
    xi = np.floor(x).astype(int) % cells
    zi = np.floor(z).astype(int) % cells
    fx = x - np.floor(x)
    fz = z - np.floor(z)
    fx = fx * fx * (3 - 2 * fx)
    fz = fz * fz * (3 - 2 * fz)
    a, b = table[zi, xi], table[zi, xi + 1]
    c, d = table[zi + 1, xi], table[zi + 1, xi + 1]
    return (a * (1 - fx) + b * fx) * (1 - fz) + (c * (1 - fx) + d * fx) * fz


# ------------------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------------------
def cornell(seed=1):
    """Room of five walls (8x8 quads each) + two boxes (4x4 quads per face): 1 024 triangles."""
    del seed  # geometry is fixed; the argument keeps the generator signature uniform
    k = 8
    walls = [
        _grid([0, 0, 0], [10, 0, 0], [0, 0, 10], k, k),      # floor
        _grid([0, 10, 0], [0, 0, 10], [10, 0, 0], k, k),     # ceiling
        _grid([0, 0, 0], [0, 10, 0], [10, 0, 0], k, k),      # back
        _grid([0, 0, 0], [0, 0, 10], [0, 10, 0], k, k),      # left
        _grid([10, 0, 0], [0, 10, 0], [0, 0, 10], k, k),     # right
    ]
    boxes = [_box([1.5, 0, 1.5], [4.5, 6, 4.5], 4), _box([5.5, 0, 5], [8.5, 3, 8], 4)]
    v, f = _merge(walls + boxes)
    return _finish("cornell", v, f, light_point=[5.0, 9.5, 5.0])


def atrium(seed=2):
    """Bumpy floor, flat ceiling with a skylight gap, 12x12 round columns, scattered crates: ~250 k."""
    rs = np.random.RandomState(seed)
    n = 200
    fv, ff = _grid([0, 0, 0], [120, 0, 0], [0, 0, 120], n, n)                      # 80 000
    fv[:, 1] = 0.35 * _value_noise(fv[:, 0] / 6.0, fv[:, 2] / 6.0, seed, 32)
    parts = [(fv, ff)]
    parts.append(_grid([0, 30, 0], [0, 0, 120], [50, 0, 0], 40, 40))                # ceiling west   3 200
    parts.append(_grid([70, 30, 0], [0, 0, 120], [50, 0, 0], 40, 40))               # ceiling east   3 200
    for ix in range(12):                                                           # 144 x 960 = 138 240
        for iz in range(12):
            parts.append(_cylinder([5 + 10 * ix, 5 + 10 * iz], 0.9 + 0.3 * rs.random_sample(), 0.0, 30.0, 24, 20))
    m = 2113                                                                        # crates  25 356
    c = np.stack([rs.random_sample(m) * 116 + 2, np.zeros(m), rs.random_sample(m) * 116 + 2], 1)
    s = 0.3 + rs.random_sample((m, 3)) * np.array([1.5, 2.5, 1.5])
    lo = c - s * np.array([0.5, 0.0, 0.5])
    parts.append(_boxes_fast(lo, lo + s))
    v, f = _merge(parts)
    sc = _finish("atrium", v, f, light_point=[60.0, 26.0, 60.0])
    # the reference's default camera (bbox.max + 2) would look at the roof from outside: stand inside
    sc.eye = np.array([6.0, 11.0, 9.0], np.float32)
    sc.target = np.array([80.0, 7.0, 95.0], np.float32)
    return sc


def _instances(unit, lo, hi):
    """One copy of the unit-cube mesh `unit` per (lo, hi) pair."""
    uv, uf = unit
    verts = (lo[:, None, :] + uv[None, :, :] * (hi - lo)[:, None, :]).reshape(-1, 3)
    faces = (uf[None, :, :] + uv.shape[0] * np.arange(lo.shape[0])[:, None, None]).reshape(-1, 3)
    return verts, faces


def _city(name, seed, buildings):
    """1000 x 1000 displaced terrain (512 x 512 quads) + `buildings` towers of 48 triangles each."""
    rs = np.random.RandomState(seed)
    n, size = 512, 1000.0

    def ground(x, z):
        return 25.0 * _value_noise(x / 100.0, z / 100.0, seed, 16) + 3.0 * _value_noise(x / 15.0, z / 15.0, seed + 1, 96)

    tv, tf = _grid([0, 0, 0], [size, 0, 0], [0, 0, size], n, n)                    # 524 288
    tv[:, 1] = ground(tv[:, 0], tv[:, 2])
    c = np.stack([rs.random_sample(buildings) * (size - 20) + 10, rs.random_sample(buildings) * (size - 20) + 10], 1)
    foot = 3.0 + rs.random_sample((buildings, 2)) * 7.0
    tall = 4.0 + rs.random_sample(buildings) ** 3 * 70.0
    base = ground(c[:, 0], c[:, 1]) - 2.0
    lo = np.stack([c[:, 0] - foot[:, 0] / 2, base, c[:, 1] - foot[:, 1] / 2], 1)
    hi = np.stack([c[:, 0] + foot[:, 0] / 2, base + tall, c[:, 1] + foot[:, 1] / 2], 1)
    v, f = _merge([(tv, tf), _instances(_box([0, 0, 0], [1, 1, 1], 2), lo, hi)])
    sc = _finish(name, v, f, light_point=[420.0, 300.0, 560.0])
    # the reference's default (bbox.max + 2 -> bbox centre) leaves half of a 4K frame as sky; come
    # closer and aim at the ground so that ~85 % of the pixels carry a geometry ray
    sc.eye = np.array([760.0, 170.0, 770.0], np.float32)
    sc.target = np.array([400.0, -120.0, 390.0], np.float32)
    return sc


def city(seed=3):
    """Displaced 512x512 terrain + 9 900 towers: 999 488 triangles (full-SAH builder branch)."""
    return _city("city", seed, 9900)


def city_big(seed=3):
    """Same with 10 625 towers: 1 034 288 triangles (> 1 000 000: median-split at the root)."""
    return _city("city_big", seed, 10625)


def _tube(p0, p1, r0, r1, sides, segments):
    """Tapered tube from p0 to p1 (any direction): sides x segments quads."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    axis = p1 - p0
    length = np.linalg.norm(axis)
    w = axis / length
    a = np.array([1.0, 0, 0]) if abs(w[0]) < 0.9 else np.array([0, 1.0, 0])
    u = np.cross(w, a)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    ang = np.linspace(0.0, 2.0 * np.pi, sides, endpoint=False)
    t = np.linspace(0.0, 1.0, segments + 1)
    rad = r0 + (r1 - r0) * t
    ring = np.cos(ang)[None, :, None] * u[None, None, :] + np.sin(ang)[None, :, None] * v[None, None, :]
    verts = (p0[None, None, :] + t[:, None, None] * axis[None, None, :] + rad[:, None, None] * ring).reshape(-1, 3)
    i = (np.arange(sides)[None, :] + sides * np.arange(segments)[:, None]).reshape(-1)
    j = (((np.arange(sides) + 1) % sides)[None, :] + sides * np.arange(segments)[:, None]).reshape(-1)
    faces = np.stack([np.stack([i, j, j + sides], 1), np.stack([i, j + sides, i + sides], 1)], 1).reshape(-1, 3)
    return verts, faces


def _arch(c0, c1, y, rise, depth, thickness, segments):
    """Semicircular arch band between two column tops c0, c1 (x,z), springing at height y: an extruded strip."""
    c0, c1 = np.asarray(c0, float), np.asarray(c1, float)
    mid, half = (c0 + c1) / 2.0, (c1 - c0) / 2.0
    span = np.linalg.norm(half)
    dirx = half / span
    nrm = np.array([-dirx[1], dirx[0]])
    ang = np.linspace(np.pi, 0.0, segments + 1)
    parts = []
    for r in (span, span - thickness):                             # outer and inner face of the band
        px = mid[None, :] + np.cos(ang)[:, None] * r * dirx[None, :]
        py = y + np.sin(ang) * r * (rise / span)
        for s in (-0.5, 0.5):
            pass
        a = np.stack([px[:, 0] - nrm[0] * depth / 2, py, px[:, 1] - nrm[1] * depth / 2], 1)
        b = np.stack([px[:, 0] + nrm[0] * depth / 2, py, px[:, 1] + nrm[1] * depth / 2], 1)
        verts = np.concatenate([a, b], 0)
        i = np.arange(segments)
        n = segments + 1
        faces = np.concatenate([np.stack([i, i + 1, i + 1 + n], 1), np.stack([i, i + 1 + n, i + n], 1)], 0)
        parts.append((verts, faces))
    return _merge(parts)


def _leaf_cards(rs, centres, radii, count, size):
    """`count` two-triangle leaf cards with random orientation, scattered in ellipsoidal crowns (centres [k,3], radii [k,3])."""
    k = rs.randint(0, centres.shape[0], count)
    d = rs.standard_normal((count, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rad = rs.random_sample(count) ** (1.0 / 3.0)
    c = centres[k] + d * rad[:, None] * radii[k]
    a = rs.standard_normal((count, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(a, rs.standard_normal((count, 3)))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    s = size * (0.6 + 0.8 * rs.random_sample(count))[:, None]
    quad = np.stack([c - a * s - b * s * 0.5, c + a * s - b * s * 0.5, c + a * s + b * s * 0.5, c - a * s + b * s * 0.5], 1)
    verts = quad.reshape(-1, 3)
    base = 4 * np.arange(count)[:, None]
    faces = np.concatenate([base + np.array([[0, 1, 2]]), base + np.array([[0, 2, 3]])], 1).reshape(-1, 3)
    return verts, faces


def courtyard(seed=4):
    """San-Miguel-class stand-in: an arcaded courtyard (two storeys of columns and arches on four sides, recessed walls,
    a tiled floor) with trees whose crowns are hundreds of thousands of randomly oriented leaf cards, thin branches,
    tables and chairs: 999 9xx triangles (full-SAH builder branch).  Shadow rays from the floor and the furniture cross
    the crowns towards a high light, so most rays walk long paths between near misses -- the tail of the frame, not
    the average, sets the time."""
    rs = np.random.RandomState(seed)
    size, storey = 60.0, 6.0
    parts = []
    fv, ff = _grid([0, 0, 0], [size, 0, 0], [0, 0, size], 150, 150)                # floor 45 000
    fv[:, 1] = 0.05 * _value_noise(fv[:, 0] / 2.0, fv[:, 2] / 2.0, seed, 48)
    parts.append((fv, ff))
    inset, ncol = 6.0, 11
    cols = np.linspace(inset, size - inset, ncol)
    ring = [(x, inset) for x in cols] + [(size - inset, z) for z in cols[1:]] + \
           [(x, size - inset) for x in cols[-2::-1]] + [(inset, z) for z in cols[-2:0:-1]]
    for level in range(2):                                                        # columns 40 x 2 x 768, arches 40 x 2 x 96
        y0 = level * storey
        for k, (x, z) in enumerate(ring):
            parts.append(_tube([x, y0, z], [x, y0 + storey * 0.7, z], 0.32, 0.26, 24, 16))
            nx, nz = ring[(k + 1) % len(ring)]
            parts.append(_arch([x, z], [nx, nz], y0 + storey * 0.7, storey * 0.28, 0.6, 0.25, 24))
        # gallery slab above the arcade, between the column ring and the outer wall
        for (o, du, dv) in (([0, y0 + storey, 0], [size, 0, 0], [0, 0, inset]), ([0, y0 + storey, size - inset], [size, 0, 0], [0, 0, inset]),
                            ([0, y0 + storey, inset], [inset, 0, 0], [0, 0, size - 2 * inset]),
                            ([size - inset, y0 + storey, inset], [inset, 0, 0], [0, 0, size - 2 * inset])):
            parts.append(_grid(o, du, dv, 40, 8))
    h = 2 * storey + 1.5                                                          # outer walls with a relief
    for (o, du, dv) in (([0, 0, 0], [size, 0, 0], [0, h, 0]), ([0, 0, size], [0, h, 0], [size, 0, 0]),
                        ([0, 0, 0], [0, h, 0], [0, 0, size]), ([size, 0, 0], [0, 0, size], [0, h, 0])):
        wv, wf = _grid(o, du, dv, 120, 28)
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        wv += n[None, :] * (0.15 * _value_noise(wv[:, 0] + wv[:, 2], wv[:, 1] * 1.7, seed + 2, 64))[:, None]
        parts.append((wv, wf))
    # trees: trunk, branches, crowns of leaf cards
    ntree = 9
    tx = np.stack([14 + 16 * (np.arange(ntree) % 3) + rs.random_sample(ntree) * 3,
                   14 + 16 * (np.arange(ntree) // 3) + rs.random_sample(ntree) * 3], 1)
    crowns_c, crowns_r = [], []
    for t in range(ntree):
        x, z = tx[t]
        top = 4.0 + rs.random_sample() * 1.5
        parts.append(_tube([x, 0, z], [x, top, z], 0.35, 0.22, 16, 12))
        for b in range(28):
            d = rs.standard_normal(3)
            d[1] = abs(d[1]) * 0.8 + 0.3
            d /= np.linalg.norm(d)
            start = np.array([x, top * (0.55 + 0.45 * rs.random_sample()), z])
            end = start + d * (2.0 + 2.5 * rs.random_sample())
            parts.append(_tube(start, end, 0.09, 0.03, 8, 6))
            crowns_c.append(end)
            crowns_r.append([1.0 + rs.random_sample(), 0.7 + 0.6 * rs.random_sample(), 1.0 + rs.random_sample()])
    # furniture: tables (5 boxes) and chairs (6 boxes)
    m = 150
    c = np.stack([rs.random_sample(m) * (size - 2 * inset - 6) + inset + 3, np.zeros(m), rs.random_sample(m) * (size - 2 * inset - 6) + inset + 3], 1)
    lo, hi = [], []
    for k in range(m):
        w, d_, hh = (1.2, 0.8, 0.75) if k % 3 == 0 else (0.45, 0.45, 0.45)
        lo.append(c[k] + [-w / 2, hh - 0.05, -d_ / 2]); hi.append(c[k] + [w / 2, hh, d_ / 2])          # top / seat
        for sx in (-1, 1):
            for sz in (-1, 1):
                p = c[k] + [sx * (w / 2 - 0.04), 0, sz * (d_ / 2 - 0.04)]
                lo.append(p - [0.03, 0, 0.03]); hi.append(p + [0.03, hh - 0.05, 0.03])
        if k % 3:
            lo.append(c[k] + [-w / 2, hh, -d_ / 2]); hi.append(c[k] + [w / 2, hh + 0.5, -d_ / 2 + 0.04])  # chair back
    parts.append(_boxes_fast(np.array(lo), np.array(hi)))
    v, f = _merge(parts)
    leaves = (999990 - f.shape[0]) // 2                                           # fill the budget with leaf cards
    parts.append(_leaf_cards(rs, np.array(crowns_c), np.array(crowns_r), leaves, 0.09))
    v, f = _merge([(v, f), parts[-1]])
    sc = _finish("courtyard", v, f, light_point=[size * 0.62, 55.0, size * 0.35])
    sc.eye = np.array([9.0, 1.7, 11.0], np.float32)                               # a visitor under the arcade corner
    sc.target = np.array([38.0, 3.2, 36.0], np.float32)
    return sc


def terrain(n=23, seed=7):
    """Small sin-free bumpy grid with many equal centroids per axis (sort-tie stress): 2*n*n tris."""
    v, f = _grid([0, 0, 0], [float(n), 0, 0], [0, 0, float(n)], n, n)
    v[:, 1] = 2.0 * _value_noise(v[:, 0] / 4.0, v[:, 2] / 4.0, seed, 8)
    return _finish(f"terrain{n}", v, f, light_point=[n * 0.5, 6.0, n * 0.5])


def calib(seed=0):
    """One far-away triangle: a frame traced against it streams the G-buffer and the mask and nothing else
    (used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE against a known byte count)."""
    del seed
    v = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]])
    sc = _finish("calib", v, np.array([[0, 1, 2]]), light_point=[0.0, 10.0, 0.0])
    sc.eye = np.array([0.0, 0.0, 0.0], np.float32)
    sc.target = np.array([0.0, 0.0, -1.0], np.float32)
    return sc


SCENES = {"cornell": cornell, "atrium": atrium, "city": city, "city_big": city_big, "calib": calib,
          "courtyard": courtyard}


# ------------------------------------------------------------------------------------------------
# scene record, OBJ text, flat expansion, camera
# ------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self, name, verts, faces, light_point):
        self.name = name
        self.verts = verts          # float32 [nv, 3]
        self.faces = faces          # uint32  [nt, 3]
        self.bbox_min = verts.min(0)
        self.bbox_max = verts.max(0)
        self.light_point = np.asarray(light_point, np.float32)
        self.light_direction = (np.ones(3) / np.sqrt(3.0)).astype(np.float32)   # cpp:245
        # camera defaults of the reference (cpp:238-242)
        self.eye = (self.bbox_max + np.float32(2.0)).astype(np.float32)
        self.target = ((self.bbox_min + self.bbox_max) * np.float32(0.5)).astype(np.float32)
        self.fovy = 1.0

    @property
    def triangle_count(self):
        return int(self.faces.shape[0])

    def flat(self):
        """loadModel's expansion (cpp:783-824): 8 floats per vertex, ``indices[i] = i``."""
        p = self.verts[self.faces.reshape(-1)]
        out = np.zeros((p.shape[0], 8), np.float32)
        out[:, :3] = p
        return out, np.arange(p.shape[0], dtype=np.uint32)

    def write_obj(self, path):
        """Real OBJ text; %.9g round-trips every float32 exactly through a correct decimal reader."""
        with open(path, "w") as fh:
            fh.write(f"# {self.name}: {self.verts.shape[0]} vertices, {self.faces.shape[0]} triangles\n")
            np.savetxt(fh, self.verts.astype(np.float64), fmt="v %.9g %.9g %.9g")
            np.savetxt(fh, self.faces.astype(np.int64) + 1, fmt="f %d %d %d")
        return path


def _finish(name, v, f, light_point):
    return Scene(name, np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.uint32), light_point)


def jitter_offsets(n, radius, seed=11):
    """n seeded offsets inside a sphere of `radius` (area-light samples for the 16-spp config)."""
    rs = np.random.RandomState(seed)
    out = np.zeros((n, 4), np.float32)
    k = 0
    while k < n:
        p = rs.random_sample(3) * 2.0 - 1.0
        if p @ p <= 1.0:
            out[k, :3] = (p * radius).astype(np.float32)
            k += 1
    return out


def cache_dir():
    d = os.environ.get("RTS_SCENE_CACHE", "/tmp/rts_scenes")
    os.makedirs(d, exist_ok=True)
    return d
