"""CPU: the OBJ reader against the reference's own zeux parser.

tests/golden/obj_cases.json holds what the REFERENCE parser (External/zeux_objparser, compiled from
its own source into oracle/_ref by oracle/Makefile) returns for a tricky OBJ text and for a list of
number spellings; when oracle/_ref is present the same comparison is also made live on a generated mesh."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from raytracedshadows_amd import api, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "obj_cases.json")))
REF = os.path.join(ROOT, "oracle", "_ref", "libzeux_objparser_ref.so")


def _expand(v, vt, vn, f):
    """loadModel's expansion (RayTracedShadows.cpp:783-824) applied to the reference parser's arrays."""
    f = np.asarray(f, np.int64).reshape(-1, 3)
    out = np.zeros((f.shape[0], 8), np.float32)
    out[:, :3] = v.reshape(-1, 3)[f[:, 0]]
    have_n = vn.size > 0
    for i, (vi, ti, ni) in enumerate(f):
        if have_n and ni >= 0:
            out[i, 3:6] = vn.reshape(-1, 3)[ni]
        if ti >= 0:
            out[i, 6:8] = vt.reshape(-1, 3)[ti, :2]
    return out


def test_number_reader_matches_reference_bits():
    for text, bits in GOLD["float_kat_bits"]:
        got, used = api.obj_parse_float(text)
        assert int(np.array([got], np.float32).view(np.uint32)[0]) == bits, text


def test_tricky_obj_matches_reference(tmp_path):
    path = str(tmp_path / "cases.obj")
    open(path, "w").write(GOLD["text"])
    verts, idx, lo, hi = api.obj_load(path)
    v = np.array(GOLD["v_bits"], np.uint32).view(np.float32)
    vt = np.array(GOLD["vt_bits"], np.uint32).view(np.float32)
    vn = np.array(GOLD["vn_bits"], np.uint32).view(np.float32)
    want = _expand(v, vt, vn, GOLD["f"])
    assert GOLD["status"] == 0
    assert verts.shape == want.shape and (idx == np.arange(want.shape[0])).all()
    assert (verts.view(np.uint32) == want.view(np.uint32)).all()       # normals present: no generation
    assert (lo == want[:, :3].min(0)).all() and (hi == want[:, :3].max(0)).all()


def test_invalid_files_are_rejected(tmp_path):
    p = str(tmp_path / "bad.obj")
    open(p, "w").write("v 0 0 0\nv 1 0 0\nf 1 2 3\n")                # index out of range -> objValidate false
    with pytest.raises(api.RtsError):
        api.obj_load(p)
    with pytest.raises(api.RtsError):
        api.obj_load(str(tmp_path / "missing.obj"))


def test_written_scene_round_trips_exactly(tmp_path):
    """%.9g text -> the reader's double-based number parser -> the same float32 bits, CRLF tolerated."""
    sc = scenes.terrain(15)
    path = sc.write_obj(str(tmp_path / "t.obj"))
    verts, idx, lo, hi = api.obj_load(path)
    flat, _ = sc.flat()
    assert (verts[:, :3].view(np.uint32) == flat[:, :3].view(np.uint32)).all()
    crlf = str(tmp_path / "t_crlf.obj")
    open(crlf, "wb").write(open(path, "rb").read().replace(b"\n", b"\r\n"))
    v2, _, _, _ = api.obj_load(crlf)
    assert (v2[:, :3].view(np.uint32) == flat[:, :3].view(np.uint32)).all()
    n = verts[:, 3:6]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)        # generated normals (cpp:826-851)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_live_against_reference_parser(tmp_path):
    ref = C.CDLL(REF)
    ref.ref_obj_parse.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint64)] * 4 + [C.c_void_p] * 4
    sc = scenes.cornell()
    path = sc.write_obj(str(tmp_path / "c.obj"))
    sizes = [C.c_uint64(0) for _ in range(4)]
    assert ref.ref_obj_parse(path.encode(), *[C.byref(s) for s in sizes], None, None, None, None) == 0
    v = np.zeros(sizes[0].value, np.float32)
    f = np.zeros(sizes[3].value, np.int32)
    ref.ref_obj_parse(path.encode(), *[C.byref(s) for s in sizes], v.ctypes.data, None, None, f.ctypes.data)
    want = v.reshape(-1, 3)[f.reshape(-1, 3)[:, 0]]
    verts, _, _, _ = api.obj_load(path)
    assert (verts[:, :3].view(np.uint32) == want.view(np.uint32)).all()
