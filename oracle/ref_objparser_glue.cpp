// TEST INFRASTRUCTURE ONLY.  C-callable glue around the REFERENCE's own OBJ parser
// (/root/reference/External/zeux_objparser/objparser.{h,cpp}), which oracle/Makefile compiles from
// where it lies into oracle/_ref/libzeux_objparser_ref.so.  Nothing of the reference is copied here:
// this file only calls its public API (objparser.h: ObjFile, objParseFile, objParseLine, objValidate).
// Used by tests/test_obj.py to pin the product's OBJ reader (raytracedshadows_amd/csrc/rts_obj.cpp).
#include "objparser.h"

#include <cstdint>
#include <cstring>

extern "C" {

// Parses `path`; returns 0 ok, 1 cannot open, 2 fails objValidate.  Sizes are element counts.
int ref_obj_parse(const char* path, uint64_t* v_size, uint64_t* vt_size, uint64_t* vn_size, uint64_t* f_size,
                  float* v, float* vt, float* vn, int* f) {
    ObjFile o;
    if (!objParseFile(o, path)) return 1;
    bool ok = objValidate(o);
    *v_size = o.v_size; *vt_size = o.vt_size; *vn_size = o.vn_size; *f_size = o.f_size;
    if (v) memcpy(v, o.v, o.v_size * sizeof(float));
    if (vt) memcpy(vt, o.vt, o.vt_size * sizeof(float));
    if (vn) memcpy(vn, o.vn, o.vn_size * sizeof(float));
    if (f) memcpy(f, o.f, o.f_size * sizeof(int));
    return ok ? 0 : 2;
}

// One "v x y z" line through objParseLine: returns the three floats the reference stores.
int ref_obj_parse_v_line(const char* line, float* xyz) {
    ObjFile o;
    objParseLine(o, line);
    if (o.v_size != 3) return 1;
    memcpy(xyz, o.v, 3 * sizeof(float));
    return 0;
}

}
