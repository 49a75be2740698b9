// Any-hit shadow-ray traversal for gfx950 (MI355X), hand-written HIP.
//
// Re-creates the work of Source/Shaders/RayTracedShadows.comp:41-151 (ray generation + bias,
// stackless miss-link traversal, slab test, Moeller-Trumbore) over the packed node stream of
// SURVEY.md Appendix A.  One shadow ray per lane, one 8x8 pixel tile per wave64 (the reference's
// local_size 8x8, comp:127).  No MFMA: the work is branchy scalar/vec3 arithmetic.
//
// Bit-exactness contract: this file is compiled with -ffp-contract=off and
// correctly rounded divide/sqrt; the slab test has an EXACT form (GLSL compare-select min/max,
// NaN-propagating) and a FAST form (v_min3/v_max3) that is only taken when no NaN can occur for any
// lane of the wave (finite origin, finite non-zero 1/d, finite BVH) -- then both forms take the same
// decision, they can differ only in the sign of a zero that is only ever compared.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rts_device.h"

namespace rts {

static constexpr uint32_t END = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------------
// scalar pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }  // GLSL min
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }  // GLSL max

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return F3{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
}
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return F3{ a.x - b.x, a.y - b.y, a.z - b.z }; }

struct Ray { F3 o; float tmax; F3 d; F3 inv; };

// comp:61-73
template <bool FAST>
__device__ __forceinline__ bool boxHit(const Ray& r, float lox, float loy, float loz, float hix, float hiy, float hiz) {
    float fx = (hix - r.o.x) * r.inv.x, fy = (hiy - r.o.y) * r.inv.y, fz = (hiz - r.o.z) * r.inv.z;
    float nx = (lox - r.o.x) * r.inv.x, ny = (loy - r.o.y) * r.inv.y, nz = (loz - r.o.z) * r.inv.z;
    if (FAST) {
        float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fmaxf(fy, ny)), __builtin_fmaxf(fz, nz));
        float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fminf(fy, ny)), __builtin_fminf(fz, nz));
        t0 = __builtin_fmaxf(t0, 0.0f);
        return t1 >= t0;
    } else {
        float tmaxx = gmax(fx, nx), tmaxy = gmax(fy, ny), tmaxz = gmax(fz, nz);
        float tminx = gmin(fx, nx), tminy = gmin(fy, ny), tminz = gmin(fz, nz);
        float t1 = gmin(tmaxx, gmin(tmaxy, tmaxz));
        float t0 = gmax(gmax(tminx, gmax(tminy, tminz)), 0.0f);
        return t1 >= t0;
    }
}

// comp:41-59
__device__ __forceinline__ bool triHit(const Ray& r, F3 v0, F3 e0, F3 e1) {
    F3 s1 = cross3(r.d, e1);
    float invd = 1.0f / dot3(s1, e0);
    F3 dd = sub3(r.o, v0);
    float b1 = dot3(dd, s1) * invd;
    F3 s2 = cross3(dd, e0);
    float b2 = dot3(r.d, s2) * invd;
    float t = dot3(e1, s2) * invd;
    if (b1 < 0.0f || b1 > 1.0f || b2 < 0.0f || b1 + b2 > 1.0f || t < 0.0f || t > r.tmax) return false;
    return true;
}

// comp:113-120
__device__ __forceinline__ float epsilonFor(float f, uint32_t diff) {
    uint32_t u = __float_as_uint(f);
    uint32_t e = (u >> 23) & 0xFFu;
    e -= (diff < e) ? diff : e;
    u = (u & ~(0xFFu << 23)) | (e << 23);
    return __uint_as_float(u);
}

__device__ __forceinline__ bool finite3(F3 v) {
    return (__builtin_fabsf(v.x) < __builtin_inff()) && (__builtin_fabsf(v.y) < __builtin_inff()) &&
           (__builtin_fabsf(v.z) < __builtin_inff());
}

// comp:128-146 (+ the point-light / multi-sample extensions documented in include/rts.h)
__device__ __forceinline__ Ray makeShadowRay(const TraceParams& p, F3 rel, uint32_t sample) {
    F3 origin{ p.cam[0] + rel.x, p.cam[1] + rel.y, p.cam[2] + rel.z };
    float mo = gmax(gmax(__builtin_fabsf(origin.x), __builtin_fabsf(origin.y)), __builtin_fabsf(origin.z));
    float mr = gmax(gmax(__builtin_fabsf(rel.x), __builtin_fabsf(rel.y)), __builtin_fabsf(rel.z));
    float bias = gmax(epsilonFor(mo, 13), epsilonFor(mr, 13));
    F3 L{ p.light[0], p.light[1], p.light[2] };
    if (p.nsamples > 1) {
        L.x = L.x + p.offsets[sample][0]; L.y = L.y + p.offsets[sample][1]; L.z = L.z + p.offsets[sample][2];
    }
    Ray r;
    if (p.lightType == 0) {
        origin.x = origin.x + L.x * bias; origin.y = origin.y + L.y * bias; origin.z = origin.z + L.z * bias;
        r.o = origin; r.tmax = 1e9f; r.d = L;
    } else {
        F3 d0 = sub3(L, origin);
        float inv = 1.0f / __builtin_sqrtf(dot3(d0, d0));
        origin.x = origin.x + (d0.x * inv) * bias; origin.y = origin.y + (d0.y * inv) * bias;
        origin.z = origin.z + (d0.z * inv) * bias;
        r.o = origin; r.tmax = 1.0f; r.d = sub3(L, origin);
    }
    r.inv = F3{ 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };   // comp:77
    return r;
}

// True when the FAST slab test is provably identical to the EXACT one for this ray.
__device__ __forceinline__ bool raySafe(const Ray& r) {
    return finite3(r.o) && finite3(r.inv) && r.inv.x != 0.0f && r.inv.y != 0.0f && r.inv.z != 0.0f;
}

// ------------------------------------------------------------------------------------------------
// node fetch: the stream is read with buffer loads (32-bit byte offsets, hardware range check)
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NodeStream {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ u32x4 vec4(uint32_t index) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(index * 16u), 0, 0);
    }
};
__device__ __forceinline__ NodeStream openStream(const TraceParams& p) {
    NodeStream s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.bvh, 0, (int)p.bvhBytes, 0x00020000);
    return s;
}
__device__ __forceinline__ F3 xyz(u32x4 v) {
    return F3{ __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z) };
}

// ------------------------------------------------------------------------------------------------
// traversal variants.  All return true when the ray is occluded.  `live` = lane owns a ray.
// ------------------------------------------------------------------------------------------------

// V_STRAIGHT: the loop exactly as the shader spells it (comp:75-111).
template <bool FAST>
__device__ __forceinline__ bool traverseStraight(const NodeStream& bvh, const Ray& r, bool live, uint32_t start = 0u) {
    uint32_t node = live ? start : END;
    while (node != END) {
        u32x4 a = bvh.vec4(node * 2), b = bvh.vec4(node * 2 + 1);
        if (a.w != END) {
            u32x4 t = bvh.vec4(a.w);
            if (triHit(r, xyz(t), xyz(a), xyz(b))) return true;
        } else if (boxHit<FAST>(r, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                                __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z))) {
            ++node;
            continue;
        }
        node = b.w;
    }
    return false;
}

// V_WHILEWHILE: descend inner nodes until every lane of the wave holds a leaf (or is done), then
// run the triangle test once for all of them.  Same set of tests per ray as the shader.
template <bool FAST>
__device__ __forceinline__ bool traverseWhileWhile(const NodeStream& bvh, const Ray& r, bool live) {
    uint32_t node = live ? 0u : END;
    for (;;) {
        u32x4 a{ 0, 0, 0, END }, b{ 0, 0, 0, END };
        while (node != END) {
            a = bvh.vec4(node * 2); b = bvh.vec4(node * 2 + 1);
            if (a.w != END) break;
            bool h = boxHit<FAST>(r, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                                  __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            node = h ? node + 1 : b.w;
        }
        if (node == END) return false;
        u32x4 t = bvh.vec4(a.w);
        if (triHit(r, xyz(t), xyz(a), xyz(b))) return true;
        node = b.w;
    }
}

// V_POSTPONE: a lane that reaches a leaf parks it (edges + v0 already fetched into registers) and
// keeps walking through the leaf's miss link; parked triangles are tested for the whole wave at
// once when some lane meets a second leaf, or when every lane has run out of nodes.  Any-hit is an
// OR over the same set of tests, so the mask is unchanged; a lane may walk a few nodes past the
// point where the shader would have returned.
template <bool FAST>
__device__ __forceinline__ bool traversePostpone(const NodeStream& bvh, const Ray& r, bool live) {
    uint32_t node = live ? 0u : END;
    bool parked = false, hit = false;
    F3 pe0{ 0, 0, 0 }, pe1{ 0, 0, 0 }, pv0{ 0, 0, 0 };
    for (;;) {
        bool conflict = false;
        u32x4 a{ 0, 0, 0, END }, b{ 0, 0, 0, END };
        if (node != END) {
            a = bvh.vec4(node * 2); b = bvh.vec4(node * 2 + 1);
            if (a.w != END) {
                if (parked) conflict = true;
                else {
                    u32x4 t = bvh.vec4(a.w);
                    pe0 = xyz(a); pe1 = xyz(b); pv0 = xyz(t);
                    parked = true;
                    node = b.w;
                }
            } else {
                bool h = boxHit<FAST>(r, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                                      __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                node = h ? node + 1 : b.w;
            }
        }
        bool walking = (node != END);
        if (__builtin_amdgcn_ballot_w64(conflict) != 0 || __builtin_amdgcn_ballot_w64(walking) == 0) {
            if (parked) {
                parked = false;
                if (triHit(r, pv0, pe0, pe1)) { hit = true; node = END; conflict = false; }
            }
            if (conflict) {      // park the leaf this lane is standing on
                u32x4 t = bvh.vec4(a.w);
                pe0 = xyz(a); pe1 = xyz(b); pv0 = xyz(t);
                parked = true;
                node = b.w;
            }
            if (__builtin_amdgcn_ballot_w64(node != END || parked) == 0) break;
        }
    }
    return hit;
}


// ------------------------------------------------------------------------------------------------
// V_PACKET: the wave walks the tree ONCE for its 64 rays.
//
// Every lane's node index only ever increases (miss links point forward) and
// every way out of a subtree leads to the same index (the subtree root's miss link).  So if `cur` is
// the smallest index any lane stands on, then after the lanes on `cur` have been tested
//     cur' = (some lane entered the subtree) ? cur + 1 : next(cur)
// is again the smallest index in the wave: the wave as a whole obeys the shader's own stackless rule
// with "box hit" = OR over its lanes.  The node is therefore wave-uniform: it is fetched with ONE
// scalar load (32 B through the scalar cache instead of 64 x 32 B through the vector L1), its fields
// sit in SGPRs, the leaf/inner branch is a scalar branch, and only lanes with idx == cur take part.
// Each lane still performs exactly the tests the shader would perform for its ray (same operands,
// same arithmetic), so the mask is bit-identical.  Measured on the headline frame an 8x8 tile visits
// 41 distinct nodes while its longest single ray visits 35 (oracle: orc_tile_union_stats).
//
// Incoherent waves (random generic rays) would visit up to 64x the nodes; the packet therefore has a
// step budget, after which its lanes are handed to the lane-per-ray loop, each from its own node.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8* ConstNodePtr;
typedef const __attribute__((address_space(4))) u32x4* ConstVec4Ptr;

// A coherent 8x8 tile needs 40-70 packet steps on the BASELINE scenes (union of its rays' paths); 64
// unrelated rays would need thousands.  After this many steps the packet dissolves and each lane goes on alone.
static constexpr int32_t PACKET_MAX_STEPS = 192;

__device__ __forceinline__ uint32_t waveMinU32(uint32_t v) {
    // smallest value over all 64 lanes (inactive-by-value lanes carry END = 0xFFFFFFFF)
    uint64_t todo = __builtin_amdgcn_ballot_w64(v != END);
    uint32_t m = END;
    while (todo) {
        int l = __builtin_ctzll(todo);
        uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        m = x < m ? x : m;
        // every lane standing on x is accounted for at once
        todo &= ~__builtin_amdgcn_ballot_w64(v == x);
    }
    return m;
}

// Packet state: `members` (wave-uniform 64-bit mask, SGPR pair) = lanes walking with the packet, i.e.
// standing on `cur`; `wait` (per lane) = the node a lane that left the packet waits on (END = nothing
// to wait for: finished or never started).  A lane leaves when its own test fails (box miss / triangle
// miss) and always waits on next(cur); the packet picks it up again when `cur` gets there.

// Hot loop, hand-written for gfx950: walk inner nodes from `cur` until the packet stands on a leaf
// (returns 1, node in n[0..7]), or runs out of nodes (cur == END) or of steps (both return 0).
// Per step: 1 scalar load (32 B), 27 VALU, 9 SALU.  The slab test is the FAST form (v_min/v_max;
// legal only when no NaN can occur, see the top of this file), operation for operation what the
// compiler emits for boxHit<true>.  Fixed scratch SGPRs s[40:52] keep the node tuple addressable.
// Hazards: the only VALU-written SGPRs (vcc from v_cmp_ge, `mem` from v_cmp_eq) are read by SALU
// instructions, which the hardware interlocks; v_cndmask reads an SALU-written mask.
__device__ __forceinline__ uint32_t packetDescendFast(const void* base, const Ray& r, uint32_t& cur, uint64_t& members,
                                                      uint32_t& wait, int32_t& budget, uint32_t (&n)[8]) {
    uint32_t leaf;
    float t0, t1, t2, t3, t4, t5, t6;
    asm volatile(
        "2:\n\t"
        "s_lshl_b32 s52, %[cur], 5\n\t"
        "s_load_dwordx8 s[40:47], %[base], s52\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cmp_lg_u32 s43, -1\n\t"
        "s_cbranch_scc1 9f\n\t"
        "v_sub_f32 %[t0], s44, %[ox]\n\t"
        "v_sub_f32 %[t1], s45, %[oy]\n\t"
        "v_sub_f32 %[t2], s46, %[oz]\n\t"
        "v_sub_f32 %[t3], s40, %[ox]\n\t"
        "v_sub_f32 %[t4], s41, %[oy]\n\t"
        "v_sub_f32 %[t5], s42, %[oz]\n\t"
        "v_mul_f32 %[t0], %[t0], %[ix]\n\t"
        "v_mul_f32 %[t1], %[t1], %[iy]\n\t"
        "v_mul_f32 %[t2], %[t2], %[iz]\n\t"
        "v_mul_f32 %[t3], %[t3], %[ix]\n\t"
        "v_mul_f32 %[t4], %[t4], %[iy]\n\t"
        "v_mul_f32 %[t5], %[t5], %[iz]\n\t"
        "v_max_f32 %[t6], %[t0], %[t3]\n\t"
        "v_min_f32 %[t0], %[t0], %[t3]\n\t"
        "v_max_f32 %[t3], %[t1], %[t4]\n\t"
        "v_min_f32 %[t1], %[t1], %[t4]\n\t"
        "v_max_f32 %[t4], %[t2], %[t5]\n\t"
        "v_min_f32 %[t2], %[t2], %[t5]\n\t"
        "v_min3_f32 %[t6], %[t6], %[t3], %[t4]\n\t"
        "v_max_f32 %[t0], %[t0], %[t1]\n\t"
        "v_max3_f32 %[t0], %[t0], %[t2], 0\n\t"
        "v_cmp_ge_f32 vcc, %[t6], %[t0]\n\t"
        "v_mov_b32 %[t1], s47\n\t"
        "s_andn2_b64 s[50:51], %[mem], vcc\n\t"
        "s_and_b64 s[48:49], %[mem], vcc\n\t"
        "v_cndmask_b32 %[wait], %[wait], %[t1], s[50:51]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_mov_b64 %[mem], s[48:49]\n\t"
        "s_add_u32 %[cur], %[cur], 1\n\t"
        "s_sub_u32 %[budget], %[budget], 1\n\t"
        "s_cbranch_scc0 2b\n\t"
        "s_branch 8f\n\t"
        "3:\n\t"
        "s_mov_b32 %[cur], s47\n\t"
        "s_cmp_eq_u32 s47, -1\n\t"
        "s_cbranch_scc1 8f\n\t"
        "v_cmp_eq_u32 %[mem], s47, %[wait]\n\t"
        "s_sub_u32 %[budget], %[budget], 1\n\t"
        "s_cbranch_scc0 2b\n\t"
        "8:\n\t"
        "s_mov_b32 %[leaf], 0\n\t"
        "s_branch 7f\n\t"
        "9:\n\t"
        "s_mov_b32 %[leaf], 1\n\t"
        "7:\n\t"
        "s_mov_b32 %[n0], s40\n\t"
        "s_mov_b32 %[n1], s41\n\t"
        "s_mov_b32 %[n2], s42\n\t"
        "s_mov_b32 %[n3], s43\n\t"
        "s_mov_b32 %[n4], s44\n\t"
        "s_mov_b32 %[n5], s45\n\t"
        "s_mov_b32 %[n6], s46\n\t"
        "s_mov_b32 %[n7], s47\n\t"
        : [cur] "+s"(cur), [mem] "+s"(members), [wait] "+v"(wait), [budget] "+s"(budget), [leaf] "=&s"(leaf),
          [n0] "=&s"(n[0]), [n1] "=&s"(n[1]), [n2] "=&s"(n[2]), [n3] "=&s"(n[3]),
          [n4] "=&s"(n[4]), [n5] "=&s"(n[5]), [n6] "=&s"(n[6]), [n7] "=&s"(n[7]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6)
        : [base] "s"(base), [ox] "v"(r.o.x), [oy] "v"(r.o.y), [oz] "v"(r.o.z),
          [ix] "v"(r.inv.x), [iy] "v"(r.inv.y), [iz] "v"(r.inv.z)
        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "vcc", "scc");
    return leaf;
}

template <bool FAST>
__device__ __forceinline__ bool traversePacket(const TraceParams& p, const NodeStream& bvh, const Ray& r, bool live) {
    const ConstNodePtr nodes = (ConstNodePtr)(uintptr_t)p.bvh;
    const ConstVec4Ptr vec4s = (ConstVec4Ptr)(uintptr_t)p.bvh;
    uint64_t members = __builtin_amdgcn_ballot_w64(live);
    if (members == 0) return false;
    uint64_t occluded = 0;               // wave-uniform mask of lanes whose ray hit a triangle
    uint32_t wait = END;
    uint32_t cur = 0;
    int32_t budget = PACKET_MAX_STEPS;
    do {                                             // single-exit loop: keeps the scalar control flow lean
        uint32_t n[8];
        bool leaf;
        if (FAST) {
            leaf = packetDescendFast(p.bvh, r, cur, members, wait, budget, n) != 0;
        } else {
            // EXACT form (NaN-propagating compare-selects), compiled: one node per trip
            const u32x8 v = nodes[cur];              // s_load_dwordx8: {a.xyz, a.w | b.xyz, b.w}
            n[0] = v.s0; n[1] = v.s1; n[2] = v.s2; n[3] = v.s3; n[4] = v.s4; n[5] = v.s5; n[6] = v.s6; n[7] = v.s7;
            leaf = n[3] != END;
            if (!leaf) {
                const bool h = boxHit<false>(r, __uint_as_float(n[0]), __uint_as_float(n[1]), __uint_as_float(n[2]),
                                             __uint_as_float(n[4]), __uint_as_float(n[5]), __uint_as_float(n[6]));
                const uint64_t in = __builtin_amdgcn_ballot_w64(h) & members;
                wait = __builtin_amdgcn_inverse_ballot_w64(members & ~in) ? n[7] : wait;
                if (in != 0) { members = in; cur = cur + 1; }
                else { cur = n[7]; members = __builtin_amdgcn_ballot_w64(wait == n[7]); }
                --budget;
            }
        }
        if (leaf) {
            const uint32_t next = n[7];
            const u32x4 t = vec4s[n[3]];
            const F3 e0{ __uint_as_float(n[0]), __uint_as_float(n[1]), __uint_as_float(n[2]) };
            const F3 e1{ __uint_as_float(n[4]), __uint_as_float(n[5]), __uint_as_float(n[6]) };
            const uint64_t hit = __builtin_amdgcn_ballot_w64(triHit(r, xyz(t), e0, e1)) & members;
            occluded |= hit;
            wait = __builtin_amdgcn_inverse_ballot_w64(members & ~hit) ? next : wait;
            wait = __builtin_amdgcn_inverse_ballot_w64(hit) ? END : wait;
            cur = next;
            members = __builtin_amdgcn_ballot_w64(wait == cur);
            if (members == 0) {                      // everyone here got occluded: jump to the lowest waiting index
                cur = waveMinU32(wait);
                members = __builtin_amdgcn_ballot_w64(wait == cur);
            }
        }
    } while (cur != END && budget >= 0);
    bool result = __builtin_amdgcn_inverse_ballot_w64(occluded);
    if (cur != END) {
        // step budget exhausted: every unfinished lane continues alone from the node it stands or waits on
        const uint32_t mine = __builtin_amdgcn_inverse_ballot_w64(members) ? cur : wait;
        const bool h = traverseStraight<FAST>(bvh, r, mine != END, mine);
        result = result || h;
    }
    return result;
}

template <int VARIANT, bool FAST>
__device__ __forceinline__ bool traverse(const TraceParams& p, const NodeStream& bvh, const Ray& r, bool live) {
    if (VARIANT == V_PACKET) return traversePacket<FAST>(p, bvh, r, live);
    if (VARIANT == V_WHILEWHILE) return traverseWhileWhile<FAST>(bvh, r, live);
    if (VARIANT == V_POSTPONE) return traversePostpone<FAST>(bvh, r, live);
    return traverseStraight<FAST>(bvh, r, live);
}

// ------------------------------------------------------------------------------------------------
// tile mapping: wave -> 8x8 pixel tile.  A 256-thread block is a 2x2 group of tiles (16x16 px).
// With swizzle on, blocks that share an XCD (blockIdx % 8, round-robin dispatch) get a contiguous
// chunk of the block sequence, and the sequence walks the image in 8-block-wide column strips so
// that co-resident blocks touch neighbouring subtrees of the BVH (per-XCD L2 locality).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool blockToXY(const TraceParams& p, uint32_t bid, uint32_t* bx, uint32_t* by) {
    uint32_t b = bid;
    if (p.swizzle) {
        uint32_t per = p.gridBlocks / 8u;           // gridBlocks is padded to a multiple of 8
        b = (bid % 8u) * per + bid / 8u;
        if (b >= p.nBlocks) return false;
        uint32_t stripBlocks = 8u * p.blocksY;
        uint32_t strip = b / stripBlocks, within = b - strip * stripBlocks;
        uint32_t width = p.blocksX - strip * 8u; if (width > 8u) width = 8u;
        *by = within / width;
        *bx = strip * 8u + (within - *by * width);
        return true;
    }
    if (b >= p.nBlocks) return false;
    *by = b / p.blocksX;
    *bx = b - *by * p.blocksX;
    return true;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void shadowMaskKernel(TraceParams p) {
    uint32_t bx, by;
    if (!blockToXY(p, blockIdx.x, &bx, &by)) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = p.rowBegin + by * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool live = (x < p.W) && (y < p.rowEnd);
    const size_t pix = (size_t)y * p.W + x;

    F3 rel{ 0.f, 0.f, 0.f };
    if (live) {
        f32x4 t = __builtin_nontemporal_load((const f32x4*)p.positions + pix);   // comp:135, read once
        rel = F3{ t.x, t.y, t.z };
    }
    const NodeStream bvh = openStream(p);
    const uint32_t ns = p.nsamples > 1 ? p.nsamples : 1u;
    uint32_t lit = 0;
    for (uint32_t s = 0; s < ns; ++s) {
        Ray r = makeShadowRay(p, rel, s);
        bool unsafe = live && !raySafe(r);
        bool occluded;
        if (p.bvhFinite && __builtin_amdgcn_ballot_w64(unsafe) == 0)
            occluded = traverse<VARIANT, true>(p, bvh, r, live);
        else
            occluded = traverse<VARIANT, false>(p, bvh, r, live);
        lit += occluded ? 0u : 1u;                                  // comp:148
    }
    if (live) __builtin_nontemporal_store((uint8_t)lit, &p.mask[pix]);   // comp:150
}

template <int VARIANT>
__global__ __launch_bounds__(256) void traceRaysKernel(TraceParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool live = i < p.nrays;
    Ray r;
    r.o = F3{ 0, 0, 0 }; r.d = F3{ 1, 1, 1 }; r.tmax = 0.f;
    if (live) {
        const float4* src = (const float4*)p.rays + i * 2;
        float4 o = src[0], d = src[1];
        r.o = F3{ o.x, o.y, o.z }; r.tmax = o.w; r.d = F3{ d.x, d.y, d.z };
    }
    r.inv = F3{ 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };
    const NodeStream bvh = openStream(p);
    bool unsafe = live && !raySafe(r);
    bool occluded;
    if (p.bvhFinite && __builtin_amdgcn_ballot_w64(unsafe) == 0)
        occluded = traverse<VARIANT, true>(p, bvh, r, live);
    else
        occluded = traverse<VARIANT, false>(p, bvh, r, live);
    if (live) p.out[i] = occluded ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from rts_api.cpp through rts_device.h)
// ------------------------------------------------------------------------------------------------
const char* kernelName(int variant, bool mask) {
    switch (variant) {
    case V_STRAIGHT: return mask ? "shadowMaskKernel<0>" : "traceRaysKernel<0>";
    case V_WHILEWHILE: return mask ? "shadowMaskKernel<1>" : "traceRaysKernel<1>";
    case V_POSTPONE: return mask ? "shadowMaskKernel<2>" : "traceRaysKernel<2>";
    case V_PACKET: return mask ? "shadowMaskKernel<3>" : "traceRaysKernel<3>";
    }
    return "?";
}

hipError_t launchShadowMask(int variant, const TraceParams& p, hipStream_t stream) {
    dim3 grid(p.gridBlocks), block(256);
    switch (variant) {
    case V_STRAIGHT: hipLaunchKernelGGL(shadowMaskKernel<V_STRAIGHT>, grid, block, 0, stream, p); break;
    case V_WHILEWHILE: hipLaunchKernelGGL(shadowMaskKernel<V_WHILEWHILE>, grid, block, 0, stream, p); break;
    case V_POSTPONE: hipLaunchKernelGGL(shadowMaskKernel<V_POSTPONE>, grid, block, 0, stream, p); break;
    case V_PACKET: hipLaunchKernelGGL(shadowMaskKernel<V_PACKET>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launchTraceRays(int variant, const TraceParams& p, hipStream_t stream) {
    dim3 grid((unsigned)((p.nrays + 255) / 256)), block(256);
    switch (variant) {
    case V_STRAIGHT: hipLaunchKernelGGL(traceRaysKernel<V_STRAIGHT>, grid, block, 0, stream, p); break;
    case V_WHILEWHILE: hipLaunchKernelGGL(traceRaysKernel<V_WHILEWHILE>, grid, block, 0, stream, p); break;
    case V_POSTPONE: hipLaunchKernelGGL(traceRaysKernel<V_POSTPONE>, grid, block, 0, stream, p); break;
    case V_PACKET: hipLaunchKernelGGL(traceRaysKernel<V_PACKET>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace rts
