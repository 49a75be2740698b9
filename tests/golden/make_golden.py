"""Regenerates the committed fixtures in this directory.  Run from the repo root:

    python tests/golden/make_golden.py

* appendix_a_4tri.json is NOT generated: it is the only recorded output of the real reference
  builder (dumped at survey time) and was typed in from there.
* cornell_128.npz / terrain_96.npz: packed BVH (product builder == oracle builder, asserted), seeded
  G-buffer positions, the oracle's mask and per-ray visit counts.  The reference itself cannot run
  here (GLSL shader, no Vulkan; BVHBuilder.cpp needs the absent librush), so these pin OUR oracle
  against regressions and the HIP kernels against the oracle -- they are not reference outputs.
* obj_cases.json: a tricky OBJ text + what the REFERENCE's own parser (oracle/_ref, built from
  /root/reference/External/zeux_objparser by oracle/Makefile) returns for it.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OBJ_TEXT = """# tricky cases for the zeux-semantics reader
v 0 0 0
v 1.5 -2.25 3e0
v +4.0e-1 .5 5.
v 1e-3 123456789.125 -0.000001
v 3.4028234e38 1e-45 1e23
v 0.1 0.2 0.3
vt 0.25 0.75
vt 1 0 0.5
vn 0 1 0
vn 0 0 -1
f 1 2 3
f 1/1/1 2/2/2 3/1/2 4/2/1
f -1 -2 -3 -4 -5
f 1//2 3//1 5//2
f 2/1 4/2 6/1
g ignored
f 1 2
f 6 5 4 3 2 1"""


def obj_cases():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzeux_objparser_ref.so"))
    path = os.path.join(HERE, "_tmp_cases.obj")
    with open(path, "w") as fh:
        fh.write(OBJ_TEXT)          # no trailing newline: the last line is the unterminated one
    sizes = [C.c_uint64(0) for _ in range(4)]
    ref.ref_obj_parse.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint64)] * 4 + [C.c_void_p] * 4
    st = ref.ref_obj_parse(path.encode(), *[C.byref(s) for s in sizes], None, None, None, None)
    v = np.zeros(sizes[0].value, np.float32)
    vt = np.zeros(sizes[1].value, np.float32)
    vn = np.zeros(sizes[2].value, np.float32)
    f = np.zeros(sizes[3].value, np.int32)
    st = ref.ref_obj_parse(path.encode(), *[C.byref(s) for s in sizes], v.ctypes.data, vt.ctypes.data,
                           vn.ctypes.data, f.ctypes.data)
    os.remove(path)
    floats = ["0", "1", "-1", "0.1", "1.5", "-2.25", "3e0", "+4.0e-1", ".5", "5.", "1e-3", "123456789.125",
              "-0.000001", "3.4028234e38", "1e-45", "1e23", "1e22", "1e-22", "1e-23", "0.30000001192092896",
              "16777217", "9007199254740993", "1.17549435e-38", "2.5E+3", "7e", "-.0", "12abc", "1e400", "1e-400"]
    ref.ref_obj_parse_v_line.argtypes = [C.c_char_p, C.c_void_p]
    kat = []
    for t in floats:
        out = np.zeros(3, np.float32)
        ref.ref_obj_parse_v_line(f"v {t} 0 0".encode(), out.ctypes.data)
        kat.append([t, int(out.view(np.uint32)[0])])
    json.dump({"text": OBJ_TEXT, "status": st, "v_bits": v.view(np.uint32).tolist(), "vt_bits": vt.view(np.uint32).tolist(),
               "vn_bits": vn.view(np.uint32).tolist(), "f": f.tolist(), "float_kat_bits": kat},
              open(os.path.join(HERE, "obj_cases.json"), "w"), indent=0)
    print("obj_cases.json:", st, len(v), len(vt), len(vn), len(f))


def masks():
    import oracle
    from raytracedshadows_amd import api, scenes
    for name, sc, W, H in [("cornell_128", scenes.cornell(), 128, 128), ("terrain_96", scenes.terrain(23), 96, 96)]:
        verts, idx = sc.flat()
        packed = oracle.bvh_build(verts, 8, idx, sc.triangle_count)
        prod = api.BVHBuilder().build(verts, 8, idx, sc.triangle_count).m_packedNodes
        assert (packed == prod).all()
        pos, hits = api.primary_positions(packed, sc.eye, sc.target, sc.fovy, W, H)
        k = api.RayTracingConstants.make(sc.eye, sc.light_direction, W, H)
        out = {"packed": packed, "positions": pos, "constants": k.as_array(), "light_point": sc.light_point}
        for tag, lt in [("dir", oracle.make_light(0, sc.light_direction)), ("point", oracle.make_light(1, sc.light_point))]:
            m, V, L, pv, pl = oracle.shadow_mask(packed, k.as_array(), lt, pos, W, H, per_ray=True)
            out[f"mask_{tag}"] = m
            out[f"visits_{tag}"] = pv.astype(np.uint16)
            out[f"leafs_{tag}"] = pl.astype(np.uint16)
            print(name, tag, "lit", float(m.mean()), "V/ray", V / (W * H), "L/ray", L / (W * H))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    obj_cases()
    masks()
