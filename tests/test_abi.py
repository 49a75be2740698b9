"""CPU: librts.so loads and exports every symbol the public headers declare (no compute calls)."""
import ctypes
import os
import re

from raytracedshadows_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtsh?_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(api.lib_path())
    names = _declared("rts.h") + _declared("rts_scene.h")
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported by librts.so: {missing}"


def test_status_strings_and_counts_without_a_gpu():
    assert api._lib.rts_status_string(0) == b"ok"
    assert b"BVH" in api._lib.rts_status_string(4)
    assert api.packed_count(1) == 3 and api.packed_count(1000) == 4998
    assert api._lib.rts_bvh_node_count(1000) == 1999


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(api.RayTracingConstants) == 64          # RayTracedShadows.h:56-62
    assert ctypes.sizeof(api.Light) == 24 + 64 * 16
    assert api.BVHNode_dtype.itemsize == 32                      # BVHBuilder.h:8-20


def test_struct_layouts_are_what_the_compiler_lays_out(tmp_path):
    """rts_constants, rts_light and rts_split_plan as gcc lays them out from include/rts.h against the ctypes mirrors the tests and
    bench.py pass to the library: size and the offset of every field (a field added to one side only would shift what follows)."""
    import subprocess
    pairs = (("rts_constants", api.RayTracingConstants), ("rts_light", api.Light), ("rts_split_plan", api.SplitPlan))
    body = ""
    for cname, mirror in pairs:
        body += f'  printf("%zu", sizeof({cname}));\n'
        body += "".join(f'  printf(" %zu", offsetof({cname}, {f}));\n' for f, _ in mirror._fields_) + '  printf("\\n");\n'
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rts.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    for (cname, mirror), line in zip(pairs, lines):
        got = [int(v) for v in line.split()]
        assert got[0] == ctypes.sizeof(mirror), cname
        assert got[1:] == [getattr(mirror, f).offset for f, _ in mirror._fields_], cname


def test_product_never_links_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", api.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "raytracedshadows_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "librts_oracle" not in text and "import oracle" not in text, f


def test_missing_library_fails_loudly_but_build_helper_still_imports():
    """The product path has no fallback: without librts.so the API import raises; the build helper does not need it."""
    import subprocess
    import sys
    env = dict(os.environ, RTS_LIB="/nonexistent/librts.so", PYTHONPATH=ROOT)
    ok = subprocess.run([sys.executable, "-c", "import raytracedshadows_amd.build, raytracedshadows_amd"], env=env,
                        capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run([sys.executable, "-c", "import raytracedshadows_amd.api"], env=env, capture_output=True, text=True)
    assert bad.returncode != 0 and "no CPU fallback" in bad.stderr
    bad = subprocess.run([sys.executable, "-c", "import raytracedshadows_amd as r; r.ShadowContext"], env=env,
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "is missing" in bad.stderr


def test_generated_asm_include_is_up_to_date(tmp_path):
    """rts_packet_asm.inc is generated: the committed file must be what tools/gen_packet_asm.py produces."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_packet_asm", os.path.join(ROOT, "tools", "gen_packet_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "out.inc")
    import sys
    argv, sys.argv = sys.argv, ["gen_packet_asm.py"]
    try:
        gen.main()
    finally:
        sys.argv = argv
    assert open(gen.OUT).read() == committed


def test_generated_wide_asm_include_is_up_to_date(tmp_path):
    """rts_wide_asm.inc is generated too: the committed file must be what tools/gen_wide_asm.py produces."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_wide_asm", os.path.join(ROOT, "tools", "gen_wide_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "out.inc")
    gen.main()
    assert open(gen.OUT).read() == committed


def test_null_arguments_are_refused_before_any_device_call():
    """Entry points check their arguments first (RTS_ERR_INVALID_ARG = 1): callable without a GPU."""
    lib = api._lib
    assert lib.rts_ctx_autotune(None, None, None, None, 0, 0, None, None, None) == 1
    assert lib.rts_ctx_set_option(None, b"kernel", 0) == 1
    assert lib.rts_ctx_set_bvh(None, None, 0) == 1
    assert lib.rts_trace_shadow_mask_device(None, None, None, None, 8, 8, 0, 8, None, None) == 1
    assert lib.rts_ctx_set_tile_order(None, None, 0) == 1


def test_plain_c_program_drives_the_host_producer(tmp_path):
    """include/*.h compile as C99 and a C program gets the Appendix-A stream out of librts.so (no GPU call)."""
    import json
    import struct
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(api.lib_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lrts",
                    "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    words = [int(w, 16) for w in out]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "appendix_a_4tri.json")))["packed"]
    assert len(words) == 4 * len(gold)
    for i, (xyz, w) in enumerate(gold):
        got = struct.unpack("<3f", struct.pack("<3I", *words[4 * i:4 * i + 3]))
        assert list(got) == [float(v) for v in xyz], (i, got, xyz)
        if w is not None:
            assert words[4 * i + 3] == int(w, 16), (i, hex(words[4 * i + 3]), w)
