// OBJ ingest with the observable behaviour of the reference's active loader:
//   parser    External/zeux_objparser/objparser.cpp:29-32 (index fix-up), 62-131 (number reader),
//             133-155 (v/vt/vn triplets), 181-269 (line dispatch + fan triangulation), 324-350 (validate)
//   expansion Source/RayTracedShadows.cpp:783-851 (flat, non-indexed Vertex stream, generated normals)
// Own implementation: the whole file is read once and scanned in place with a cursor; faces are
// triangulated straight into the flat vertex stream's index triples.  The number reader's arithmetic
// (digits accumulated in a double, one scaling by an exact power of ten) is what fixes the vertex
// bits, so it is kept operation-for-operation and pinned against the reference parser built from
// its own source (tests/test_obj.py, oracle/_ref).
#include "../../include/rts_scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline bool isDigit(char c) { return (unsigned)(c - '0') < 10u; }
inline const char* skipBlanks(const char* s) { while (*s == ' ' || *s == '\t') ++s; return s; }

// objparser.cpp:62-131
float readNumber(const char* s, const char** end) {
    static const double pow10[] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
    const int kPow = (int)(sizeof(pow10) / sizeof(pow10[0]));
    s = skipBlanks(s);
    double sign = 1;
    if (*s == '-') { sign = -1; ++s; } else if (*s == '+') ++s;
    double mant = 0;
    int p10 = 0;
    for (; isDigit(*s); ++s) mant = mant * 10 + (double)(*s - '0');
    if (*s == '.')
        for (++s; isDigit(*s); ++s) { mant = mant * 10 + (double)(*s - '0'); --p10; }
    if ((*s | ' ') == 'e') {
        ++s;
        int esign = 1;
        if (*s == '-') { esign = -1; ++s; } else if (*s == '+') ++s;
        int e = 0;
        // (saturated: the reference accumulates in an int with no limit, which is undefined behaviour beyond 2^31;
        //  any |exponent| >= 100000 gives the same 0 / inf as a larger one)
        for (; isDigit(*s); ++s) if (e < 100000) e = e * 10 + (*s - '0');
        p10 += esign * e;
    }
    *end = s;
    if ((unsigned)(-p10) < (unsigned)kPow) return (float)(sign * mant / pow10[-p10]);
    if ((unsigned)p10 < (unsigned)kPow) return (float)(sign * mant * pow10[p10]);
    return (float)(sign * mant * std::pow(10.0, p10));
}

// objparser.cpp:34-60
int readInt(const char* s, const char** end) {
    s = skipBlanks(s);
    bool neg = (*s == '-');
    if (*s == '-' || *s == '+') ++s;
    unsigned v = 0;
    for (; isDigit(*s); ++s) if (v < 0x0CCCCCCCu) v = v * 10 + (unsigned)(*s - '0');   // saturates below 2^31
    *end = s;
    return neg ? -(int)v : (int)v;
}

struct Corner { int v, vt, vn; };

struct Mesh {
    std::vector<float> v, vt, vn;      // stride 3 each (vt keeps uvw, objparser.h:11)
    std::vector<Corner> corners;       // 3 per triangle
};

inline int fixIndex(int idx, size_t count) { return idx >= 0 ? idx - 1 : (int)count + idx; } // cpp:29-32

void readTriple(std::vector<float>& dst, const char* s) {
    for (int k = 0; k < 3; ++k) dst.push_back(readNumber(s, &s));
}

void readFace(Mesh& m, const char* s) {     // cpp:228-268
    const size_t nv = m.v.size() / 3, nvt = m.vt.size() / 3, nvn = m.vn.size() / 3;
    Corner first{ 0, 0, 0 }, prev{ 0, 0, 0 };
    int have = 0;
    while (*s) {
        int vi = 0, vti = 0, vni = 0;
        s = skipBlanks(s);
        vi = readInt(s, &s);                 // cpp:133-155
        if (*s == '/') {
            ++s;
            if (*s != '/') vti = readInt(s, &s);
            if (*s == '/') { ++s; vni = readInt(s, &s); }
        }
        if (vi == 0) break;
        Corner c{ fixIndex(vi, nv), fixIndex(vti, nvt), fixIndex(vni, nvn) };
        if (have == 0) { first = c; have = 1; }
        else if (have == 1) { prev = c; have = 2; }
        else {
            m.corners.push_back(first); m.corners.push_back(prev); m.corners.push_back(c);
            prev = c;
        }
    }
}

void readLine(Mesh& m, const char* l) {     // cpp:181-269
    if (l[0] == 'v' && l[1] == ' ') readTriple(m.v, l + 2);
    else if (l[0] == 'v' && l[1] == 't' && l[2] == ' ') readTriple(m.vt, l + 3);
    else if (l[0] == 'v' && l[1] == 'n' && l[2] == ' ') readTriple(m.vn, l + 3);
    else if (l[0] == 'f' && l[1] == ' ') readFace(m, l + 2);
}

bool validate(const Mesh& m) {              // cpp:324-350
    const size_t nv = m.v.size() / 3, nvt = m.vt.size() / 3, nvn = m.vn.size() / 3;
    for (const Corner& c : m.corners) {
        if (c.v < 0 || (size_t)c.v >= nv) return false;
        if (c.vt >= 0 && (size_t)c.vt >= nvt) return false;
        if (c.vn >= 0 && (size_t)c.vn >= nvn) return false;
    }
    return true;
}

bool parseFile(const char* path, Mesh& m) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    size_t pos = 0;
    while (pos < text.size()) {          // every '\n'-terminated line, then the unterminated rest
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        else text[eol] = '\0';
        readLine(m, text.c_str() + pos);
        pos = eol + 1;
    }
    return true;
}

} // namespace

extern "C" float rtsh_obj_parse_float(const char* text, int* consumed) {
    const char* end = text;
    float v = readNumber(text, &end);
    if (consumed) *consumed = (int)(end - text);
    return v;
}

extern "C" int rtsh_obj_load(const char* path, float* vertices, size_t cap, uint32_t* vertex_count,
                             float bbox_min[3], float bbox_max[3]) {
    if (!path || !vertex_count) return RTS_ERR_INVALID_ARG;
    Mesh m;
    try {                                            // bad_alloc / length_error must not cross the C ABI
        if (!parseFile(path, m) || !validate(m)) return RTS_ERR_INVALID_ARG;
    } catch (...) {
        return RTS_ERR_CAPACITY;
    }
    const size_t nverts = m.corners.size();
    if (nverts > 0xFFFFFFFFull) return RTS_ERR_CAPACITY;
    *vertex_count = (uint32_t)nverts;
    if (!vertices) return RTS_OK;
    if (cap < nverts) return RTS_ERR_CAPACITY;
    const bool haveNormals = !m.vn.empty();
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < nverts; ++i) {           // RayTracedShadows.cpp:791-824
        const Corner& c = m.corners[i];
        float* o = vertices + i * 8;
        for (int k = 0; k < 3; ++k) {
            o[k] = m.v[(size_t)c.v * 3 + k];
            o[3 + k] = (haveNormals && c.vn >= 0) ? m.vn[(size_t)c.vn * 3 + k] : 0.0f;
            if (o[k] < lo[k]) lo[k] = o[k];
            if (o[k] > hi[k]) hi[k] = o[k];
        }
        o[6] = c.vt >= 0 ? m.vt[(size_t)c.vt * 3 + 0] : 0.0f;
        o[7] = c.vt >= 0 ? m.vt[(size_t)c.vt * 3 + 1] : 0.0f;
    }
    if (!haveNormals) {                              // RayTracedShadows.cpp:826-851 (shading only)
        for (size_t t = 0; t + 2 < nverts; t += 3) {
            float* a = vertices + t * 8; float* b = a + 8; float* c = b + 8;
            float u[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, w[3] = { c[0] - b[0], c[1] - b[1], c[2] - b[2] };
            float n[3] = { u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0] };
            float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (len > 0) for (int k = 0; k < 3; ++k) n[k] /= len;
            for (int k = 0; k < 3; ++k) { a[3 + k] += n[k]; b[3 + k] += n[k]; c[3 + k] += n[k]; }
        }
        for (size_t i = 0; i < nverts; ++i) {
            float* n = vertices + i * 8 + 3;
            float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (len > 0) for (int k = 0; k < 3; ++k) n[k] /= len;
        }
    }
    if (bbox_min) for (int k = 0; k < 3; ++k) bbox_min[k] = lo[k];
    if (bbox_max) for (int k = 0; k < 3; ++k) bbox_max[k] = hi[k];
    return RTS_OK;
}
