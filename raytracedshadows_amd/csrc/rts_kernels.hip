// Any-hit shadow-ray traversal for gfx950 (MI355X), hand-written HIP.
//
// Re-creates the work of Source/Shaders/RayTracedShadows.comp:41-151 (ray generation + bias,
// stackless miss-link traversal, slab test, Moeller-Trumbore) over the packed node stream of
// SURVEY.md Appendix A.  One shadow ray per lane, one 8x8 pixel tile per wave64 (the reference's
// local_size 8x8, comp:127).  No MFMA: the work is branchy scalar/vec3 arithmetic.
//
// Two families of kernels: lane-per-ray (each lane walks its own ray with vector loads: `straight`,
// `while-while`, `postpone`, `share`) and PACKET (the wave walks the union of its rays' paths, the node
// is wave-uniform and fetched with one scalar load; descent + leaf test hand-written in gfx950
// assembly, rts_packet_asm.inc).  DESIGN.md section 4 has the measurements behind every choice.
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

// 1.0f / x, correctly rounded, in three instructions where that is proven: for 2^-100 <= |x| <= 2^100, v_rcp_f32 followed by one
// Newton step in fma arithmetic IS the IEEE quotient on gfx950 -- checked for every one of the 2^32 bit patterns against the
// division (rts_selftest_reciprocal below, run by the -m gpu suite; Markstein's theorem covers all but the all-ones mantissas,
// the exhaustive run covers those too; the range keeps every intermediate normal).  Callers decide per WAVE (one ballot) and
// keep the general division -- eleven instructions -- for waves with a value outside the range (0, denormal, huge, Inf, NaN).
__device__ __forceinline__ bool rcpInRange(float x) {
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;               // biased exponent: 27 .. 227 <=> 2^-100 <= |x| < 2^101
    return e - 27u <= 199u;                                               // ... cut at 2^100 exclusive of the last binade: 27 .. 226
}
__device__ __forceinline__ float rcpFast(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// comp:41-59
__device__ __forceinline__ bool triHit(const Ray& r, F3 v0, F3 e0, F3 e1) {
    F3 s1 = cross3(r.d, e1);
    const float det = dot3(s1, e0);
    float invd;                                                          // = 1.0f / det (comp:44), bit for bit
    if (__builtin_amdgcn_ballot_w64(!rcpInRange(det)) == 0) invd = rcpFast(det); else invd = 1.0f / det;
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
// rts_light.table (include/rts.h): where in the offset table the pixel starts
__device__ __forceinline__ uint32_t hash32(uint32_t v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
__device__ __forceinline__ uint32_t sampleIndex(const TraceParams& p, uint32_t sample, uint32_t pixel) {
    if (p.lightTable == 0) return sample;
    const uint32_t j = __umulhi(hash32(pixel + p.pixelBase), p.lightTable) + sample;   // start < table, sample < nsamples <= table
    return j >= p.lightTable ? j - p.lightTable : j;
}

__device__ __forceinline__ Ray makeShadowRay(const TraceParams& p, F3 rel, uint32_t sample, uint32_t pixel = 0) {
    F3 origin{ p.cam[0] + rel.x, p.cam[1] + rel.y, p.cam[2] + rel.z };
    float mo = gmax(gmax(__builtin_fabsf(origin.x), __builtin_fabsf(origin.y)), __builtin_fabsf(origin.z));
    float mr = gmax(gmax(__builtin_fabsf(rel.x), __builtin_fabsf(rel.y)), __builtin_fabsf(rel.z));
    float bias = gmax(epsilonFor(mo, 13), epsilonFor(mr, 13));
    F3 L{ p.light[0], p.light[1], p.light[2] };
    if (p.nsamples > 1) {
        const uint32_t j = sampleIndex(p, sample, pixel);
        L.x = L.x + p.offsets[j][0]; L.y = L.y + p.offsets[j][1]; L.z = L.z + p.offsets[j][2];
    }
    Ray r;
    if (p.lightType == 0) {
        origin.x = origin.x + L.x * bias; origin.y = origin.y + L.y * bias; origin.z = origin.z + L.z * bias;
        r.o = origin; r.tmax = 1e9f; r.d = L;
    } else {
        F3 d0 = sub3(L, origin);
        const float len = __builtin_sqrtf(dot3(d0, d0));
        // (lanes without a pixel carry texel 0 or zeros: they share the decision, which is only about speed)
        float inv;
        if (__builtin_amdgcn_ballot_w64(!rcpInRange(len)) == 0) inv = rcpFast(len); else inv = 1.0f / len;
        origin.x = origin.x + (d0.x * inv) * bias; origin.y = origin.y + (d0.y * inv) * bias;
        origin.z = origin.z + (d0.z * inv) * bias;
        r.o = origin; r.tmax = 1.0f; r.d = sub3(L, origin);
    }
    if (__builtin_amdgcn_ballot_w64(!(rcpInRange(r.d.x) && rcpInRange(r.d.y) && rcpInRange(r.d.z))) == 0)
        r.inv = F3{ rcpFast(r.d.x), rcpFast(r.d.y), rcpFast(r.d.z) };
    else
        r.inv = F3{ 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };   // comp:77
    return r;
}

// True when the FAST slab test is provably identical to the EXACT one for this ray.
__device__ __forceinline__ bool raySafe(const Ray& r) {
    // v_cmp_class masks: 0x1F8 = any finite value, 0x198 = finite and not zero (+-normal, +-denormal)
    return __builtin_amdgcn_classf(r.o.x, 0x1F8) && __builtin_amdgcn_classf(r.o.y, 0x1F8) && __builtin_amdgcn_classf(r.o.z, 0x1F8) &&
           __builtin_amdgcn_classf(r.inv.x, 0x198) && __builtin_amdgcn_classf(r.inv.y, 0x198) && __builtin_amdgcn_classf(r.inv.z, 0x198);
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

// ------------------------------------------------------------------------------------------------
// V_SHARE: lane-per-ray walk with WORK SHARING inside the wave (used for dissolved packets too).
//
// A stackless walk that stands on node c and may go up to `bound` visits, in index order, subtree(c)
// and then everything from next(c) on.  The two parts are independent: any lane can walk
// [next(c), bound) for the same ray while the original lane keeps [c, next(c)) -- and since any-hit is an
// OR over the tests, the pieces can run in parallel on different lanes.  Whenever at least SHARE_MIN_IDLE
// lanes have nothing to do, the k-th idle lane takes the second part of the k-th busy lane's range (ray
// copied with ds_bpermute, lane numbers exchanged through 256 B of LDS).  A piece that finds a hit marks
// its owner in a wave-uniform mask; pieces of an occluded owner stop.  The set of tests per ray is a
// superset of the shader's up to its first hit, so the mask is unchanged; the long tail of a wave whose
// rays scatter (atrium: one ray visiting 381 nodes at L2 latency) is spread over all 64 lanes.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t SHARE_MIN_IDLE = 8;
static constexpr uint32_t OOB_VEC4 = 0x0FFFFFF0u;      // vec4 index whose byte offset (0xFFFFFF00) lies beyond any accepted stream

__device__ __forceinline__ uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// What the PIECES of one split tile share (see traversePiece): two words of device memory touched with relaxed agent-scope
// atomics only (the pieces of a tile are one-wave workgroups that may run on different XCDs).  Nobody ever waits on them:
// state[0] collects the lanes (= rays of the tile) some piece found occluded -- a piece asks for it every few steps and drops
// those rays, the request travelling while the walk goes on --, state[1] counts the pieces that are done.
struct PieceShare {
    uint64_t* state = nullptr;
    uint64_t seen = 0;              // state[0] as of the previous request
    uint32_t* log = nullptr;        // split planning only: visit log of this piece {count, node indices ...}
    uint32_t logCap = 0, logCount = 0;
    bool diag = false;              // "piece_stats": where a piece's time goes
    uint64_t tPacketEnd = 0, tLaneStart = 0;
    uint32_t nodes = 0, entries = 0;
    // An agent-scope round trip takes 1-2 us (profiles/r04/piece_stats_*.log), several steps of a walk: the answer to the
    // PREVIOUS request is looked at, then the next request is issued into the same registers (the asm ties the address to
    // the look, so that the compiler neither hoists the load above it nor copies -- and therefore waits for -- its result).
    __device__ __forceinline__ uint64_t poll() {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seen);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seen >> 32));
        // (a GLOBAL load: a flat one would also count as an LDS / scalar-memory operation and hold up the node fetches)
        typedef __attribute__((address_space(1))) uint64_t* GlobalU64;
        uint64_t addr = (uint64_t)(uintptr_t)state;
        asm volatile("" : "+s"(addr) : "s"(lo), "s"(hi));
        seen = __hip_atomic_load((GlobalU64)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ void publish(uint64_t lanes) {
        if (laneId() == 0) (void)__hip_atomic_fetch_or(state, lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

struct ShareDiag {              // diagnostics ("wave_stats"): iterations of the loop and active lanes summed over them
    bool on = false;
    uint32_t iterations = 0, laneSteps = 0;
    uint64_t tDissolve = 0;     // clock when the (last) packet of this wave dissolved
};

// CONFIRM (rays handed over by the wide packet walk, which culls with a conservative test): such a ray can stand inside a
// subtree whose root box it does not hit by the exact test, and reach a leaf there through a miss link.  A triangle hit
// then only counts if the ray hits the box of the leaf's PARENT by the exact test -- by the enclosure property that is
// "the reference's walk reaches this leaf" (rts_wide.hip).  The parent's index comes from the private parent table.
// TEAM (a piece of a split tile): the walk ends at `bound0`, owners that another piece found occluded are dropped (polled every
// fourth iteration), owners found occluded here are published.
template <bool FAST, bool CONFIRM = false, bool TEAM = false>
__device__ __forceinline__ bool traverseShare(const NodeStream& bvh, Ray r, bool live, uint32_t start, uint32_t* ldsSlots,
                                              ShareDiag* diag = nullptr, const uint32_t* parents = nullptr, uint32_t bound0 = END,
                                              PieceShare* team = nullptr) {
    uint32_t node = live ? start : END, bound = bound0, owner = laneId();
    uint64_t occludedOwners = 0;                       // wave-uniform
    uint32_t iter = 0;
    // The loop is rotated: the node of the NEXT iteration is requested before the triangle of this one is tested.
    // A leaf's successor is known without any arithmetic (its miss link), and the triangle's v0 is a load that
    // depends on the node just fetched; waiting for it before moving on costs a second memory latency in every
    // iteration in which any lane stands on a leaf.  Here v0 and the next node travel together (v0 first: loads
    // return in order), and the triangle test runs while the next node is still on its way.
    // All three loads of an iteration are issued unconditionally (a lane with nothing to fetch asks for an index
    // beyond the buffer: the range check answers 0 without touching memory).  With a load inside a divergent branch
    // the compiler can only wait for "all outstanding loads" at the first use, which would serialise them again.
    bool active = node < bound;
    u32x4 a = bvh.vec4(active ? node * 2u : OOB_VEC4), b = bvh.vec4(active ? node * 2u + 1u : OOB_VEC4);
    for (;;) {
        if constexpr (TEAM) {
            // (node fetches and this request return in order: every sixteenth iteration may wait for a round trip)
            if ((iter & 15u) == 0) {
                occludedOwners |= team->poll();
                active = active && !((occludedOwners >> owner) & 1ull);
            }
        }
        const uint64_t act = __builtin_amdgcn_ballot_w64(active);
        if (act == 0) break;
        if (diag && diag->on) { diag->iterations += 1u; diag->laneSteps += (uint32_t)__builtin_popcountll(act); }
        if constexpr (TEAM) {
            if (team->log) {                                          // planning: a quarter of the visits, rotating over the lanes
                const bool sel = active && (((laneId() + iter) & 3u) == 0);
                const uint64_t m = __builtin_amdgcn_ballot_w64(sel);
                const uint32_t at = team->logCount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (sel && at < team->logCap) team->log[1u + at] = node;
                team->logCount += (uint32_t)__builtin_popcountll(m);
            }
        }
        uint32_t next = node;                                        // lanes that do not move keep their (finished) range
        if ((iter++ & 3u) == 0) {
            const uint64_t idle = ~act;
            const bool canGive = active && b.w < bound;              // there is a second part to give away
            const uint64_t givers = __builtin_amdgcn_ballot_w64(canGive);
            const uint32_t nIdle = (uint32_t)__builtin_popcountll(idle), nGive = (uint32_t)__builtin_popcountll(givers);
            if (nIdle >= SHARE_MIN_IDLE && nGive != 0) {
                const uint32_t pairs = nIdle < nGive ? nIdle : nGive;
                const uint32_t lane = laneId();                      // (recomputed: cheaper than a register held across the walk)
                const uint32_t below = (1u << (lane & 31u)) - 1u;
                const uint32_t rankG = lane < 32 ? __builtin_popcount((uint32_t)givers & below)
                                                 : __builtin_popcount((uint32_t)givers) + __builtin_popcount((uint32_t)(givers >> 32) & below);
                const uint32_t rankI = lane < 32 ? __builtin_popcount((uint32_t)idle & below)
                                                 : __builtin_popcount((uint32_t)idle) + __builtin_popcount((uint32_t)(idle >> 32) & below);
                const bool gives = canGive && rankG < pairs;
                const bool takes = !active && rankI < pairs;
                if (gives) ldsSlots[rankG] = lane;                   // k-th giver announces itself ...
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t src = takes ? ldsSlots[rankI] : lane;  // ... to the k-th idle lane
                // the taker copies the ray and the second part of the range (every lane reads its `src`, givers read themselves)
                const int sel = (int)(src << 2);
                const float ox = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.o.x)));
                const float oy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.o.y)));
                const float oz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.o.z)));
                const float dx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.d.x)));
                const float dy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.d.y)));
                const float dz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.d.z)));
                const float ix = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.inv.x)));
                const float iy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.inv.y)));
                const float iz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.inv.z)));
                const float tm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, r.tmax)));
                const uint32_t srcNext = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)b.w);
                const uint32_t srcBound = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)bound);
                const uint32_t srcOwner = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)owner);
                if (takes) {
                    r.o = F3{ ox, oy, oz }; r.d = F3{ dx, dy, dz }; r.inv = F3{ ix, iy, iz }; r.tmax = tm;
                    next = srcNext; bound = srcBound; owner = srcOwner;   // its first node is requested below
                }
                if (gives) bound = b.w;                              // keeps [node, next(node))
                __builtin_amdgcn_wave_barrier();
            }
        }
        const bool leaf = active && a.w != END;
        const u32x4 v0 = bvh.vec4(leaf ? a.w : OOB_VEC4);
        if (active) {
            if (leaf) {
                next = b.w;
            } else {
                const bool h = boxHit<FAST>(r, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                                            __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                next = h ? node + 1 : b.w;
            }
        }
        // request the next node, then test the triangle while it travels
        const bool nextActive = next < bound;
        const uint32_t nv = nextActive ? next * 2u : OOB_VEC4;
        const u32x4 na = bvh.vec4(nv), nb = bvh.vec4(nv + 1u);
        bool hitNow = leaf && triHit(r, xyz(v0), xyz(a), xyz(b));
        if (CONFIRM && hitNow) {
            const uint32_t parent = parents[node];
            const u32x4 pa = bvh.vec4(parent * 2u), pb = bvh.vec4(parent * 2u + 1u);
            hitNow = boxHit<FAST>(r, __uint_as_float(pa.x), __uint_as_float(pa.y), __uint_as_float(pa.z),
                                  __uint_as_float(pb.x), __uint_as_float(pb.y), __uint_as_float(pb.z));
        }
        uint64_t hits = __builtin_amdgcn_ballot_w64(hitNow);
        if (TEAM && hits) {
            uint64_t fresh = 0;
            for (uint64_t hm = hits; hm; hm &= hm - 1) fresh |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)owner, __builtin_ctzll(hm));
            team->publish(fresh);
        }
        while (hits) {                                               // rare: record the owners that just got occluded
            const int l = __builtin_ctzll(hits);
            occludedOwners |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)owner, l);
            hits &= hits - 1;
        }
        node = next; a = na; b = nb;
        // pieces of an owner that is known to be occluded have nothing left to prove
        active = nextActive && !((occludedOwners >> owner) & 1ull);
    }
    return (occludedOwners >> laneId()) & 1ull;
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
// Incoherent waves (random generic rays) would visit up to 64x the nodes; the packet therefore checks
// its own coherence every few side-steps and, when it no longer pays, dissolves: its lanes are handed
// to the lane-per-ray loop (with work sharing), each from the node it stands or waits on.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8* ConstNodePtr;
typedef const __attribute__((address_space(4))) u32x4* ConstVec4Ptr;

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

// Packet state: `members[k]` (wave-uniform 64-bit masks, SGPR pairs) = lanes whose ray of set k walks
// with the packet, i.e. stands on `cur`; `wait[k]` (per lane) = the node a ray that left the packet
// waits on (END = nothing to wait for: finished or never started).  A ray leaves when its own test
// fails (box miss / triangle miss) and always waits on next(cur); the packet picks it up again when
// `cur` gets there.  A lane carries K rays (K = 1, 2, 4: the wave is an 8x8, 16x8, 16x16 pixel tile):
// the union of the paths grows slowly with the tile (41 / 48 / 53 nodes on the headline frame), so
// more rays per lane means fewer dependent scalar loads per ray at the same VALU work.
//
// The descent over inner nodes is hand-written gfx950 assembly (rts_packet_asm.inc, generated by
// tools/gen_packet_asm.py): per step one s_load_dwordx8, K slab tests, ~6+5K scalar ops.
// Slab test, "ordered" form: for a lane with 1/d.x > 0,  f = (hi.x-o.x)*inv.x >= n = (lo.x-o.x)*inv.x
// because lo.x <= hi.x and IEEE subtraction, multiplication by a positive number and rounding are all
// monotone; so the shader's tmax.x = max(f,n) IS f and tmin.x = min(f,n) IS n (equal values are
// interchangeable), and the other way round for 1/d.x < 0.  When every ray of the wave has the same
// sign pattern the choice of plane is a compile-time operand swap (8 instantiations) and the six
// per-axis min/max disappear: t1 = min3(far), t0 = max(max3(near), 0), hit = t1 >= t0 -- the same
// values the FAST/EXACT forms compare.  Preconditions: no NaN can occur (FAST rule), every inner node
// has bboxMin <= bboxMax (checked at upload), uniform sign pattern; otherwise the generic FAST form
// (v_min/v_max per axis) runs, and waves with an unsafe ray go lane-per-ray with the EXACT form.
#include "rts_packet_asm.inc"

template <int K, bool PREFETCH = false>
__device__ __forceinline__ void traversePacket(const TraceParams& p, const NodeStream& bvh, const Ray (&r)[K],
                                               const bool (&live)[K], bool (&result)[K], uint32_t* lds,
                                               int32_t* sideStepsLeft = nullptr, ShareDiag* shareDiag = nullptr) {
    // (the stream's address as an explicitly wave-uniform value: when this function is inlined into a loop over tiles the
    //  compiler may keep the kernel argument in VGPRs, which the asm's scalar loads cannot take)
    const uint64_t bvhAddr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)(uintptr_t)p.bvh >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p.bvh);
    const void* const bvhBase = (const void*)(uintptr_t)bvhAddr;
    const ConstNodePtr nodes = (ConstNodePtr)(uintptr_t)bvhAddr;
    const ConstVec4Ptr vec4s = (ConstVec4Ptr)(uintptr_t)bvhAddr;
    uint64_t members[K], occluded[K];
    uint32_t wait[K];
    uint64_t any = 0, unsafe = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        members[k] = __builtin_amdgcn_ballot_w64(live[k]);
        any |= members[k];
        unsafe |= __builtin_amdgcn_ballot_w64(live[k] && !raySafe(r[k]));
        occluded[k] = 0;
        wait[k] = END;
        result[k] = false;
    }
    if (any == 0) return;
    if (!p.bvhFinite || unsafe != 0) {       // a NaN could occur somewhere in this wave: EXACT form, lane per ray
#pragma unroll
        for (int k = 0; k < K; ++k) result[k] = traverseShare<false>(bvh, r[k], live[k], 0u, lds);
        return;
    }
    // sign pattern of 1/d over all live rays of the wave: uniform -> ordered slab test.  (1/d is finite and non-zero for
    // every live ray here, so "negative" is the sign bit: three shifts per ray instead of three compares + ballots.)
    uint32_t form = 8;
    if (p.bvhOrdered) {
        uint32_t first = 0;
        uint64_t differs = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t oct = (__float_as_uint(r[k].inv.x) >> 31) | ((__float_as_uint(r[k].inv.y) >> 31) << 1) |
                                 ((__float_as_uint(r[k].inv.z) >> 31) << 2);
            if (k == 0) first = (uint32_t)__builtin_amdgcn_readlane((int)oct, __builtin_ctzll(any));   // `any` != 0 here ...
            differs |= __builtin_amdgcn_ballot_w64(oct != first) & members[k];
        }
        // ... but its lowest lane need not be live in set 0 when K > 1: then `first` may belong to a dead ray, which only
        // costs the uniform form (a wrong `first` never matches every live ray unless it is their common pattern)
        if (differs == 0 && (K == 1 || (members[0] >> __builtin_ctzll(any)) & 1ull)) form = first;
    }
    form = (uint32_t)__builtin_amdgcn_readfirstlane((int)form);
    // Dissolve rule (evaluated inside the asm loop).  A packet step serves the rays standing on `cur`; a
    // lane-per-ray step serves every unfinished ray of the wave but costs about twice as much.  Every
    // p.packetBudget side-steps the rays picked up per side-step are compared with the rays still alive:
    // below p.packetShare/16 of them it is cheaper to let every ray continue alone (atrium: 92 % of the rays
    // die early and the rest scatter between the columns).
    uint32_t cur = 0;
    const uint32_t window = p.packetBudget - 1u;
    const uint32_t thr = p.packetBudget * p.packetShare;
    int32_t budget = (int32_t)window;
    uint32_t acc = 0;
    bool leaf;
    do {
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        budget = __builtin_amdgcn_readfirstlane(budget);
        acc = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc);
#pragma unroll
        for (int k = 0; k < K; ++k)
            members[k] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(members[k] >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)members[k]);
        if constexpr (K == 1 && !PREFETCH) {
            occluded[0] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(occluded[0] >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)occluded[0]);
            // leaves are handled inside the asm loop; it only comes back when the packet is finished (cur == END),
            // dissolves, or (code 2) stands on a node nobody waits on after all its rays got occluded
            const uint32_t code = packetDescendLeaf(form, bvhBase, r, cur, members, wait, occluded, budget, acc, thr, window);
            if (code == 2) {
                cur = waveMinU32(wait[0]);
                members[0] = __builtin_amdgcn_ballot_w64(wait[0] == cur);
                leaf = true;                         // (keeps the loop going; the leaf block below is skipped)
                continue;
            }
            leaf = false;
        } else if constexpr (PREFETCH && K == 1)
            leaf = packetDescendPrefetch(form, bvhBase, r, cur, members, wait, budget, acc, thr, window) != 0;
        else
            leaf = packetDescend(form, bvhBase, r, cur, members, wait, budget, acc, thr, window) != 0;
        if (leaf) {
            // the packet stands on a leaf: one triangle, tested by the rays that are here
            const u32x8 n = nodes[cur];
            const u32x4 t = vec4s[n.s3];
            const uint32_t next = n.s7;
            const F3 e0{ __uint_as_float(n.s0), __uint_as_float(n.s1), __uint_as_float(n.s2) };
            const F3 e1{ __uint_as_float(n.s4), __uint_as_float(n.s5), __uint_as_float(n.s6) };
            uint64_t here = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (members[k] != 0) {
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(triHit(r[k], xyz(t), e0, e1)) & members[k];
                    occluded[k] |= hit;
                    wait[k] = __builtin_amdgcn_inverse_ballot_w64(members[k] & ~hit) ? next : wait[k];
                    wait[k] = __builtin_amdgcn_inverse_ballot_w64(hit) ? END : wait[k];
                }
                members[k] = __builtin_amdgcn_ballot_w64(wait[k] == next);
                here |= members[k];
            }
            cur = next;
            if (here == 0) {                 // every ray that stood here got occluded: go to the lowest waiting node
                uint32_t lowest = END;
#pragma unroll
                for (int k = 0; k < K; ++k) { const uint32_t m = waveMinU32(wait[k]); lowest = m < lowest ? m : lowest; }
                cur = lowest;
#pragma unroll
                for (int k = 0; k < K; ++k) members[k] = __builtin_amdgcn_ballot_w64(wait[k] == cur);
            }
        }
    } while (leaf && cur != END);
    const bool dissolve = cur != END;
#pragma unroll
    for (int k = 0; k < K; ++k) result[k] = __builtin_amdgcn_inverse_ballot_w64(occluded[k]);
    if (sideStepsLeft) *sideStepsLeft = dissolve ? -1 : 0;
    if (cur != END) {
        // dissolved (not coherent enough for a packet): every unfinished ray continues alone
        if (shareDiag && shareDiag->on) shareDiag->tDissolve = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t mine = __builtin_amdgcn_inverse_ballot_w64(members[k]) ? cur : wait[k];
            const bool h = traverseShare<true>(bvh, r[k], mine != END, mine, lds, shareDiag);
            result[k] = result[k] || h;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// V_WIDE: the packet walks the private WIDE nodes (rts_wide.hip) with a stack of (node, member mask).
//
// One dependent fetch (128 B, two s_load_dwordx16) brings the boxes of the four nodes two levels down; every member ray
// tests all four, the packet continues with the first slot somebody hit and pushes the others.  The stack lives in three
// VGPRs, entry i in lane i (v_readlane / v_writelane with a scalar lane index); the masks on it are exact membership, so no
// per-lane bookkeeping ("who waits where") is left.
//
// What is tested, and why the mask is still the reference's: by the enclosure property (see rts_wide.hip) the reference's
// ray is occluded iff there is a leaf whose triangle it hits AND whose parent's box it hits (comp:61-73, exact
// arithmetic).  Anything that only DECIDES WHICH leaves to look at may be any superset test.  So inner culling uses the
// cheap form  t = fma(plane, 1/d, -(o/d -+ slack))  (10 VALU per box instead of 16: one fma per plane, the slack folded
// into two per-ray constants) and only a triangle HIT is confirmed with the exact slab test of the slot's box (a leaf
// slot carries its parent's box).  The slack covers the two roundings of the exact form and the one of the fma for every
// plane inside the root box (derivation: DESIGN.md 4.7; checked exhaustively on the CPU by orc_wide_packet_sim: no exact
// hit is ever missed).
//
// Dissolve: every `packetBudget` nodes the members per node are compared with the rays alive; below packetShare/16 (or
// when the stack is nearly full) every unfinished ray continues the reference's stackless walk from the node of its
// topmost stack entry, lane per ray with work sharing, confirming triangle hits (traverseShare<.., CONFIRM>).
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) u32x16* ConstWidePtr;
typedef const __attribute__((address_space(4))) uint32_t* ConstU32Ptr;

struct WideRay { F3 cU, cD; };                     // o/d - slack (for upper bounds), o/d + slack (for lower bounds)

__device__ __forceinline__ uint64_t uniform64(const void* ptr) {
    const uint64_t v = (uint64_t)(uintptr_t)ptr;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// slack >= 3.5u E |1/d| + 2u |o/d| + tiny, E = largest |plane - o| over the root box, u = 2^-24.  False when a product
// could overflow (the wave then takes the exact lane-per-ray walk).
__device__ __forceinline__ bool wideRaySetup(const Ray& r, const float* rootLo, const float* rootHi, WideRay& w) {
    const float o[3] = { r.o.x, r.o.y, r.o.z }, inv[3] = { r.inv.x, r.inv.y, r.inv.z };
    float cU[3], cD[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float E = __builtin_fmaxf(__builtin_fabsf(rootLo[a] - o[a]), __builtin_fabsf(rootHi[a] - o[a]));
        const float oi = o[a] * inv[a];
        const float slack = (E * __builtin_fabsf(inv[a])) * 4.76837158e-7f + __builtin_fabsf(oi) * 2.38418579e-7f + 7.5e-37f;
        const float M = __builtin_fmaxf(__builtin_fabsf(rootLo[a]), __builtin_fabsf(rootHi[a])) * __builtin_fabsf(inv[a]);
        cU[a] = oi - slack; cD[a] = oi + slack;
        ok = ok && (slack < __builtin_inff()) && (M < 1e37f) && (__builtin_fabsf(oi) < 1e37f);
    }
    w.cU = F3{ cU[0], cU[1], cU[2] }; w.cD = F3{ cD[0], cD[1], cD[2] };
    return ok;
}

// OCT 0..7: bit a set <=> 1/d component a is negative in every lane (far plane = bboxMin on that axis); 8: per lane.
template <int OCT>
__device__ __forceinline__ bool cheapBox(const float* lo, const float* hi, const F3& inv, const WideRay& w) {
    float fx, fy, fz, nx, ny, nz;
    if (OCT < 8) {
        fx = __builtin_fmaf((OCT & 1) ? lo[0] : hi[0], inv.x, -w.cU.x); nx = __builtin_fmaf((OCT & 1) ? hi[0] : lo[0], inv.x, -w.cD.x);
        fy = __builtin_fmaf((OCT & 2) ? lo[1] : hi[1], inv.y, -w.cU.y); ny = __builtin_fmaf((OCT & 2) ? hi[1] : lo[1], inv.y, -w.cD.y);
        fz = __builtin_fmaf((OCT & 4) ? lo[2] : hi[2], inv.z, -w.cU.z); nz = __builtin_fmaf((OCT & 4) ? hi[2] : lo[2], inv.z, -w.cD.z);
    } else {
        const bool sx = inv.x < 0.f, sy = inv.y < 0.f, sz = inv.z < 0.f;
        fx = __builtin_fmaf(sx ? lo[0] : hi[0], inv.x, -w.cU.x); nx = __builtin_fmaf(sx ? hi[0] : lo[0], inv.x, -w.cD.x);
        fy = __builtin_fmaf(sy ? lo[1] : hi[1], inv.y, -w.cU.y); ny = __builtin_fmaf(sy ? hi[1] : lo[1], inv.y, -w.cD.y);
        fz = __builtin_fmaf(sz ? lo[2] : hi[2], inv.z, -w.cU.z); nz = __builtin_fmaf(sz ? hi[2] : lo[2], inv.z, -w.cD.z);
    }
    const float t1 = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), 0.0f);
    return t1 >= t0;
}


// ------------------------------------------------------------------------------------------------
// Lane-per-ray walk over the WIDE nodes (what a dissolved wide packet continues in).
//
// The stackless lane-per-ray walk (traverseShare) pays one dependent fetch per node, and in a dissolved wave that fetch
// usually misses the L2 (0.5 - 0.9 us per iteration measured: a wave waits for its slowest lane).  Here a lane fetches a
// wide node (its own: 7 x 16 B) and decides four boxes per dependent fetch, so a ray needs about a third of the
// iterations.  Every lane keeps its own stack of pending nodes in LDS (LANE_STACK entries, entry e of lane l at word
// e * 64 + l: conflict-free); a push that does not fit is not lost: the node's index in the stream is remembered
// (lowest one per lane) and those lanes finish with the stackless walk from there (any-hit is an OR over a superset of
// tests; hits are confirmed against the leaf's parent box as everywhere in the wide kernels).
// Slab test: the cheap form with the plane picked per lane by the sign of 1/d (16 VALU per box).
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t LANE_STACK = 16;

__device__ __forceinline__ bool cheapBoxLane(const float* lo, const float* hi, const F3& inv, const WideRay& w) {
    const bool sx = inv.x < 0.f, sy = inv.y < 0.f, sz = inv.z < 0.f;
    const float fx = __builtin_fmaf(sx ? lo[0] : hi[0], inv.x, -w.cU.x), nx = __builtin_fmaf(sx ? hi[0] : lo[0], inv.x, -w.cD.x);
    const float fy = __builtin_fmaf(sy ? lo[1] : hi[1], inv.y, -w.cU.y), ny = __builtin_fmaf(sy ? hi[1] : lo[1], inv.y, -w.cD.y);
    const float fz = __builtin_fmaf(sz ? lo[2] : hi[2], inv.z, -w.cU.z), nz = __builtin_fmaf(sz ? hi[2] : lo[2], inv.z, -w.cD.z);
    const float t1 = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), 0.0f);
    return t1 >= t0;
}

struct LaneWalk {
    __amdgpu_buffer_rsrc_t priv;   // the private copy: wide nodes, then the triangle records (one allocation)
    uint32_t trisOff;              // byte offset of the triangle records in it
    uint32_t* stack;               // this wave's LANE_STACK * 64 words of LDS
    uint32_t sp = 0;               // entries on this lane's stack
    uint32_t lostNode = END, lostLeaf = END;   // lowest refs among the items that did not fit on it (refs grow with the stream index)

    // byte offset (in the private copy) of what an item refers to: a wide node, or (bit 0 set) a triangle record
    __device__ __forceinline__ uint32_t offsetOf(uint32_t ref) const { return (ref & 1u) ? trisOff + (ref - 1u) : ref; }
    __device__ __forceinline__ void push(uint32_t ref, bool doIt) {
        const bool fits = doIt && sp < LANE_STACK;
        if (fits) { stack[sp * 64u + laneId()] = ref; ++sp; }
        if (doIt && !fits) {                                    // remember where the stackless walk has to start (no fetch here:
            if (ref & 1u) lostLeaf = ref < lostLeaf ? ref : lostLeaf;   //  wide nodes and triangle records are numbered in stream order)
            else lostNode = ref < lostNode ? ref : lostNode;
        }
    }
    // stream index of the lowest item that was lost (END: none): dword 28 of a wide node, dword 9 of a triangle record
    __device__ __forceinline__ uint32_t lostIndex() const {
        const uint32_t a = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(priv, (int)(lostNode != END ? lostNode + 112u : 0xFFFFFF00u), 0, 0);
        const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(priv, (int)(lostLeaf != END ? offsetOf(lostLeaf) + 36u : 0xFFFFFF00u), 0, 0);
        const uint32_t ia = lostNode != END ? a : END, ib = lostLeaf != END ? b : END;
        return ia < ib ? ia : ib;
    }
    __device__ __forceinline__ uint32_t pop(bool doIt) {
        uint32_t v = END;
        if (doIt && sp > 0) { --sp; v = stack[sp * 64u + laneId()]; }
        return v;
    }
};

// `cur`: the item this lane takes up first (a wide node, or a triangle record with bit 0 set), END = nothing; pending
// items are on the lane's stack.  One iteration = one item per lane = ONE memory latency: seven 16-byte loads from the
// item's offset (a node uses all of them, a triangle record the first three), then either four box tests (every slot
// somebody hit becomes an item: the first is taken up next, the others are pushed) or one triangle test.
__device__ __forceinline__ bool laneWideWalk(LaneWalk& lw, const Ray& r, const WideRay& w, uint32_t cur, ShareDiag* diag) {
    bool occluded = false;
    for (;;) {
        const bool active = cur != END;
        const uint64_t act = __builtin_amdgcn_ballot_w64(active);
        if (act == 0) break;
        if (diag && diag->on) { diag->iterations += 1u; diag->laneSteps += (uint32_t)__builtin_popcountll(act); }
        const bool isLeaf = active && (cur & 1u);
        const bool isNode = active && !(cur & 1u);
        const uint32_t off = active ? lw.offsetOf(cur) : 0xFFFFFF00u;      // beyond the buffer: zeros, no memory access
        u32x4 q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(lw.priv, (int)(off + 16u * i), 0, 0);
        // ---- a triangle (comp:41-59), then -- only on a hit -- the exact slab test of its leaf's parent box (comp:61-73)
        if (__builtin_amdgcn_ballot_w64(isLeaf) != 0) {
            const F3 v0{ __uint_as_float(q[0].x), __uint_as_float(q[0].y), __uint_as_float(q[0].z) };
            const F3 e0{ __uint_as_float(q[0].w), __uint_as_float(q[1].x), __uint_as_float(q[1].y) };
            const F3 e1{ __uint_as_float(q[1].z), __uint_as_float(q[1].w), __uint_as_float(q[2].x) };
            bool t = isLeaf && triHit(r, v0, e0, e1);
            if (__builtin_amdgcn_ballot_w64(t) != 0)                    // the parent's box travels in the record (dwords 10-15)
                t = t && boxHit<true>(r, __uint_as_float(q[2].z), __uint_as_float(q[2].w), __uint_as_float(q[3].x),
                                      __uint_as_float(q[3].y), __uint_as_float(q[3].z), __uint_as_float(q[3].w));
            occluded = occluded || t;
        }
        // ---- a wide node: four cheap box tests; every slot hit is an item
        uint32_t next = END;
        if (__builtin_amdgcn_ballot_w64(isNode) != 0) {
            float pl[24];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                pl[4 * i + 0] = __uint_as_float(q[i].x); pl[4 * i + 1] = __uint_as_float(q[i].y);
                pl[4 * i + 2] = __uint_as_float(q[i].z); pl[4 * i + 3] = __uint_as_float(q[i].w);
            }
            const uint32_t ref[4] = { q[6].x, q[6].y, q[6].z, q[6].w };
#pragma unroll
            for (int k = 3; k >= 0; --k) {                           // last slot first: the first one ends up as `next`
                const bool hit = isNode && ref[k] != END && cheapBoxLane(&pl[6 * k], &pl[6 * k + 3], r.inv, w);
                lw.push(next, hit && next != END);
                next = hit ? ref[k] : next;
            }
        }
        if (occluded) lw.sp = 0;                                     // nothing left to prove for this ray
        const uint32_t popped = lw.pop(active && !occluded && next == END);
        cur = (!active || occluded) ? END : (next != END ? next : popped);
    }
    return occluded;
}

// A dissolved wide packet: every pending (node, members) is on the stack.  A ray continues the reference's stackless walk
// (lane per ray, work sharing, triangle hits confirmed) from the LOWEST node index among the entries it is a member of:
// everything it still has to look at lies at or after that node in the stream.
__device__ __forceinline__ bool wideDissolve(const TraceParams& p, const NodeStream& bvh, const Ray& r, const WideRay& w, uint64_t occ,
                                             uint32_t sp, uint32_t stRef, uint32_t stLo, uint32_t stHi, uint32_t* lds,
                                             uint32_t* laneStack, ShareDiag* diag) {
    const uint64_t wideAddr = uniform64(p.wide);
    if (diag && diag->on) diag->tDissolve = __builtin_amdgcn_s_memtime();
    auto entryMask = [&](uint32_t e) {
        return ((((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)stHi, (int)e)) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)stLo, (int)e)) & ~occ;
    };
    if (p.wideLane && laneStack) {
        // every ray takes its pending nodes onto a stack of its own and walks the wide nodes alone
        LaneWalk lw;
        lw.priv = __builtin_amdgcn_make_buffer_rsrc((void*)p.wide, 0, (int)p.wideBytes, 0x00020000);
        lw.trisOff = p.trisOffset;
        lw.stack = laneStack;
        for (uint32_t e = 0; e < sp; ++e) {
            const uint64_t m = entryMask(e);
            if (m == 0) continue;
            const uint32_t eref = (uint32_t)__builtin_amdgcn_readlane((int)stRef, (int)e);
            lw.push(eref, __builtin_amdgcn_inverse_ballot_w64(m));
        }
        const uint32_t first = lw.pop(true);
        bool h = laneWideWalk(lw, r, w, first, diag);
        // rays whose stack overflowed finish with the stackless walk from the lowest node they could not keep
        if (__builtin_amdgcn_ballot_w64((lw.lostNode != END || lw.lostLeaf != END) && !h) != 0) {
            const uint32_t from = h ? END : lw.lostIndex();
            h = traverseShare<true, true>(bvh, r, from != END, from, lds, diag, p.parents) || h;
        }
        return __builtin_amdgcn_inverse_ballot_w64(occ) || h;
    }
    uint32_t start = END;
    for (uint32_t e = 0; e < sp; ++e) {
        const uint64_t m = entryMask(e);
        if (m == 0) continue;
        const uint32_t eref = (uint32_t)__builtin_amdgcn_readlane((int)stRef, (int)e);
        const uint32_t self = *(ConstU32Ptr)(uintptr_t)(wideAddr + eref + 112u);
        if (__builtin_amdgcn_inverse_ballot_w64(m)) start = self < start ? self : start;
    }
    const bool h = traverseShare<true, true>(bvh, r, start != END, start, lds, diag, p.parents);
    return __builtin_amdgcn_inverse_ballot_w64(occ) || h;
}

#include "rts_wide_asm.inc"

// entries on the stack when a node is taken up: it pushes at most 3 and a dissolve one more, and lane 63 of the stack
// registers is not the stack's in the assembly form (it keeps EXEC) -- the same limit in both forms
static constexpr uint32_t WIDE_STACK_LIMIT = 59;

template <int OCT>
__device__ __forceinline__ bool wideWalk(const TraceParams& p, const NodeStream& bvh, const Ray& r, const WideRay& w,
                                         uint64_t liveMask, uint32_t* lds, uint32_t* laneStack, int32_t* dissolved, ShareDiag* diag) {
    const uint64_t wideAddr = uniform64(p.wide), triAddr = uniform64(p.tris);
    uint32_t stRef = 0, stLo = 0, stHi = 0;            // the stack: entry i lives in lane i
    uint32_t sp = 0;
    uint64_t occ = 0;
    uint32_t curRef = 0;
    uint64_t curM = liveMask;
    bool have = true, dissolve = false;
    const uint32_t window = p.packetBudget, thr = p.packetBudget * p.packetShare;
    uint32_t acc = 0, left = window;
    auto push = [&](uint32_t ref, uint64_t m) {
        // (lane select through M0: a VOP3 instruction may read one SGPR, and the data is one already)
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                     : "+v"(stRef), "+v"(stLo), "+v"(stHi)
                     : "s"(ref), "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "s"(sp)
                     : "m0");
        ++sp;
    };
    auto entryMask = [&](uint32_t e) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)stHi, (int)e) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)stLo, (int)e);
    };
    for (;;) {
        if (!have) {
            if (sp == 0) break;
            --sp;
            curRef = (uint32_t)__builtin_amdgcn_readlane((int)stRef, (int)sp);
            curM = entryMask(sp) & ~occ;
            if (curM == 0) continue;
        }
        have = false;
        acc += (uint32_t)__builtin_popcountll(curM);
        if (--left == 0) {
            const uint32_t alive = (uint32_t)__builtin_popcountll(liveMask & ~occ);
            dissolve = acc * 16u < alive * thr;
            acc = 0; left = window;
        }
        if (dissolve || sp > WIDE_STACK_LIMIT) { push(curRef, curM); dissolve = true; break; }
        const ConstWidePtr np = (ConstWidePtr)(uintptr_t)(wideAddr + curRef);
        const u32x16 n0 = np[0], n1 = np[1];
        float pl[24];
#pragma unroll
        for (int d = 0; d < 16; ++d) pl[d] = __uint_as_float(n0[d]);
#pragma unroll
        for (int d = 0; d < 8; ++d) pl[16 + d] = __uint_as_float(n1[d]);
        const uint32_t ref[4] = { n1[8], n1[9], n1[10], n1[11] };
        uint64_t h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = __builtin_amdgcn_ballot_w64(cheapBox<OCT>(&pl[6 * k], &pl[6 * k + 3], r.inv, w)) & curM;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                   // leaf slots somebody hit: one triangle each
            if (h[k] == 0 || !(ref[k] & 1u) || ref[k] == END) continue;
            const ConstVec4Ptr tp = (ConstVec4Ptr)(uintptr_t)(triAddr + (ref[k] - 1u));     // 64-byte record
            const u32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const F3 v0{ __uint_as_float(t0.x), __uint_as_float(t0.y), __uint_as_float(t0.z) };
            const F3 e0{ __uint_as_float(t0.w), __uint_as_float(t1.x), __uint_as_float(t1.y) };
            const F3 e1{ __uint_as_float(t1.z), __uint_as_float(t1.w), __uint_as_float(t2.x) };
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(h[k] & ~occ);
            bool t = mine && triHit(r, v0, e0, e1);
            if (__builtin_amdgcn_ballot_w64(t) != 0) {
                // the hit counts iff the reference's walk reaches this leaf: exact slab test of its parent's box
                t = t && boxHit<true>(r, pl[6 * k], pl[6 * k + 1], pl[6 * k + 2], pl[6 * k + 3], pl[6 * k + 4], pl[6 * k + 5]);
                occ |= __builtin_amdgcn_ballot_w64(t);
            }
        }
        uint32_t candRef = 0;
        uint64_t candM = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) {                                  // inner slots, last first: the first ends up as `cur`
            const uint64_t m = h[k] & ~occ;
            if (m == 0 || (ref[k] & 1u)) continue;
            if (candM != 0) push(candRef, candM);
            candRef = ref[k]; candM = m;
        }
        if (candM != 0) { curRef = candRef; curM = candM; have = true; }
    }
    if (dissolved) *dissolved = dissolve ? -1 : 0;
    if (!dissolve) return __builtin_amdgcn_inverse_ballot_w64(occ);
    return wideDissolve(p, bvh, r, w, occ, sp, stRef, stLo, stHi, lds, laneStack, diag);
}

template <bool ASM>
__device__ __forceinline__ bool traverseWide(const TraceParams& p, const NodeStream& bvh, const Ray& r, bool live, uint32_t* lds,
                                             uint32_t* laneStack, int32_t* dissolved, ShareDiag* diag) {
    const uint64_t liveMask = __builtin_amdgcn_ballot_w64(live);
    if (dissolved) *dissolved = 0;
    if (liveMask == 0) return false;
    const u32x8 root = *(ConstNodePtr)(uintptr_t)uniform64(p.bvh);          // node 0: the root's box
    const float rootLo[3] = { __uint_as_float(root.s0), __uint_as_float(root.s1), __uint_as_float(root.s2) };
    const float rootHi[3] = { __uint_as_float(root.s4), __uint_as_float(root.s5), __uint_as_float(root.s6) };
    WideRay w;
    const bool ok = wideRaySetup(r, rootLo, rootHi, w) && raySafe(r);
    if (__builtin_amdgcn_ballot_w64(live && !ok) != 0)                       // a NaN or an overflow could occur: exact walk
        return traverseShare<false>(bvh, r, live, 0u, lds);
    const uint32_t oct = (__float_as_uint(r.inv.x) >> 31) | ((__float_as_uint(r.inv.y) >> 31) << 1) | ((__float_as_uint(r.inv.z) >> 31) << 2);
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)oct, __builtin_ctzll(liveMask));
    const uint32_t form = (uint32_t)__builtin_amdgcn_readfirstlane((__builtin_amdgcn_ballot_w64(live && oct != first) == 0) ? (int)first : 8);
    if constexpr (ASM) {
        // the loop in gfx950 assembly (rts_wide_asm.inc, tools/gen_wide_asm.py); the C++ form below is the same algorithm
        uint32_t stRef = 0, stLo = 0, stHi = 0, sp = 0;
        uint64_t occ = 0;
        const uint64_t liveU = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(liveMask >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)liveMask);
        // (tmax is a property of the light, not of the ray: comp:145 / the point-light extension)
        const float tmaxUniform = p.lightType == 0 ? 1e9f : 1.0f;
        const uint32_t st = wideDescend(form, (const void*)(uintptr_t)uniform64(p.wide), (const void*)(uintptr_t)uniform64(p.tris), r, w,
                                        tmaxUniform, liveU, occ, sp, stRef, stLo, stHi, p.packetBudget, p.packetBudget * p.packetShare);
        if (dissolved) *dissolved = st ? -1 : 0;
        if (st == 0) return __builtin_amdgcn_inverse_ballot_w64(occ);
        return wideDissolve(p, bvh, r, w, occ, sp, stRef, stLo, stHi, lds, laneStack, diag);
    } else {
        switch (form) {
        case 0: return wideWalk<0>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 1: return wideWalk<1>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 2: return wideWalk<2>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 3: return wideWalk<3>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 4: return wideWalk<4>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 5: return wideWalk<5>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 6: return wideWalk<6>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        case 7: return wideWalk<7>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        default: return wideWalk<8>(p, bvh, r, w, liveMask, lds, laneStack, dissolved, diag);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SPLIT TILES: a tile that was measured to be long (rts_ctx_plan_splits) is walked by S one-wave workgroups, the PIECES.
// Every piece sets up all 64 rays of the tile and walks ONE index range [a, b) of the node stream; the ranges partition
// [0, N) at the quantiles of the visit log of a planning walk, so that each holds about 1/S of the tile's work.
//
// Why any cut is exact.  Wide nodes are numbered in stream order and a subtree is a contiguous index range, so "slot k's
// subtree is [index of slot k, index of slot k + 1)" (rts_wide.hip, dwords 28..31) lets the walk skip every subtree that lies
// outside [a, b); what it enters is culled with the conservative test as everywhere in the wide kernels, a triangle is tested
// by the piece whose range holds its leaf (a neighbour may test it again: any-hit is an OR), and a triangle HIT counts iff
// the ray hits the box of the leaf's parent by the exact test -- which by the enclosure property is "the reference's walk
// reaches this leaf" (rts_wide.hip).  The OR over the pieces is therefore the reference's any-hit (comp:75-111).  After a
// dissolve each ray continues the stackless walk from max(its lowest pending node, a) up to b, hits confirmed the same way.
// A wave with a ray the conservative test is not proven for (NaN / overflow possible) is walked whole, exactly, by piece 0.
//
// What the pieces share: PieceShare (two words of device memory, nobody waits).  The piece that finishes last stores the
// tile's 64 bytes and leaves the two words zero for the next launch.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wideWalkRange(const TraceParams& p, const NodeStream& bvh, const Ray& r, const WideRay& w, uint64_t liveMask,
                                                  uint32_t a, uint32_t b, uint32_t entryRef, PieceShare& team, uint32_t* lds) {
    const uint64_t wideAddr = uniform64(p.wide), triAddr = uniform64(p.tris);
    uint32_t stRef = 0, stLo = 0, stHi = 0;            // the stack: entry i lives in lane i (as in wideWalk)
    uint32_t sp = 0;
    uint64_t occ = 0;
    uint32_t curRef = entryRef;                        // the lowest wide node whose subtree holds [a, b): the levels above it only cull
    uint64_t curM = liveMask;
    bool have = true, dissolve = false;
    const uint32_t window = p.packetBudget, thr = p.packetBudget * p.packetShare;
    uint32_t acc = 0, left = window, pollLeft = 1;
    auto push = [&](uint32_t ref, uint64_t m) {
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                     : "+v"(stRef), "+v"(stLo), "+v"(stHi)
                     : "s"(ref), "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "s"(sp)
                     : "m0");
        ++sp;
    };
    auto entryMask = [&](uint32_t e) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)stHi, (int)e) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)stLo, (int)e);
    };
    if (!team.log) {
        // The everyday piece: the loop in assembly (rts_wide_asm.inc: wideDescendRange), four pops at a time; in between, a
        // look at what the other pieces found (the request of the previous look), and every fourth time the dissolve rule.
        push(entryRef, liveMask);
        const float tmaxUniform = p.lightType == 0 ? 1e9f : 1.0f;
        const void* const wb = (const void*)(uintptr_t)wideAddr;
        const void* const tb = (const void*)(uintptr_t)triAddr;
        const uint32_t thr16 = 16u * p.packetShare;
        uint32_t turns = 0;
        uint64_t told = 0;                                              // what this piece has published
        for (;;) {
            const uint32_t st = wideDescendRange(wb, tb, r, w, tmaxUniform, a, b, occ, sp, stRef, stLo, stHi, acc, 4u);
            if (team.diag) team.nodes += 4u;
            if (occ & ~told) { team.publish(occ & ~told); told = occ; }
            if (st == 0) break;
            if (st == 1) { dissolve = true; break; }
            occ |= team.poll();
            told |= occ;
            if ((liveMask & ~occ) == 0) { sp = 0; break; }
            if (++turns == 4u) {
                const uint32_t alive = (uint32_t)__builtin_popcountll(liveMask & ~occ);
                if (acc * 16u < alive * thr16) { dissolve = true; break; }
                acc = 0; turns = 0;
            }
        }
    } else
    for (;;) {
        if (!have) {
            if (sp == 0) break;
            --sp;
            curRef = (uint32_t)__builtin_amdgcn_readlane((int)stRef, (int)sp);
            curM = entryMask(sp) & ~occ;
            if (curM == 0) continue;
        }
        have = false;
        if (--pollLeft == 0) {                                          // rays another piece found occluded (as of eight nodes ago)
            pollLeft = 8;
            occ |= team.poll();
            if ((liveMask & ~occ) == 0) { sp = 0; break; }
            curM &= ~occ;
            if (curM == 0) continue;
        }
        acc += (uint32_t)__builtin_popcountll(curM);
        if (--left == 0) {
            const uint32_t alive = (uint32_t)__builtin_popcountll(liveMask & ~occ);
            dissolve = acc * 16u < alive * thr;
            acc = 0; left = window;
        }
        if (dissolve || sp > WIDE_STACK_LIMIT) { push(curRef, curM); dissolve = true; break; }
        const ConstWidePtr np = (ConstWidePtr)(uintptr_t)(wideAddr + curRef);
        const u32x16 n0 = np[0], n1 = np[1];
        float pl[24];
#pragma unroll
        for (int d = 0; d < 16; ++d) pl[d] = __uint_as_float(n0[d]);
#pragma unroll
        for (int d = 0; d < 8; ++d) pl[16 + d] = __uint_as_float(n1[d]);
        const uint32_t ref[4] = { n1[8], n1[9], n1[10], n1[11] };
        const uint32_t self = n1[12];
        const uint32_t idx[4] = { self + 1u, n1[13], n1[14], n1[15] };   // first node of slot k's subtree (slot 0: a lower bound)
        if (team.diag) ++team.nodes;
        if (team.log) {                                                   // planning: a packet node weighs eight lane visits
            const uint32_t at = team.logCount + laneId();
            if (laneId() < 8u && at < team.logCap) team.log[1u + at] = self;
            team.logCount += 8u;
        }
        uint64_t h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool in;
            if (ref[k] == END) in = false;
            else if (ref[k] & 1u) in = k == 0 ? (self + 2u >= a && self + 1u < b) : (idx[k] >= a && idx[k] < b);
            else in = idx[k] < b && (k == 3 || ref[k == 3 ? 3 : k + 1] == END || idx[k == 3 ? 3 : k + 1] > a);
            h[k] = in ? (__builtin_amdgcn_ballot_w64(cheapBox<8>(&pl[6 * k], &pl[6 * k + 3], r.inv, w)) & curM) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                   // leaf slots somebody hit: one triangle each
            if (h[k] == 0 || !(ref[k] & 1u)) continue;
            const ConstVec4Ptr tp = (ConstVec4Ptr)(uintptr_t)(triAddr + (ref[k] - 1u));
            const u32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const F3 v0{ __uint_as_float(t0.x), __uint_as_float(t0.y), __uint_as_float(t0.z) };
            const F3 e0{ __uint_as_float(t0.w), __uint_as_float(t1.x), __uint_as_float(t1.y) };
            const F3 e1{ __uint_as_float(t1.z), __uint_as_float(t1.w), __uint_as_float(t2.x) };
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(h[k] & ~occ);
            bool t = mine && triHit(r, v0, e0, e1);
            if (__builtin_amdgcn_ballot_w64(t) != 0) {
                t = t && boxHit<true>(r, pl[6 * k], pl[6 * k + 1], pl[6 * k + 2], pl[6 * k + 3], pl[6 * k + 4], pl[6 * k + 5]);
                const uint64_t fresh = __builtin_amdgcn_ballot_w64(t);
                if (fresh) { occ |= fresh; team.publish(fresh); }
            }
        }
        uint32_t candRef = 0;
        uint64_t candM = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const uint64_t m = h[k] & ~occ;
            if (m == 0 || (ref[k] & 1u)) continue;
            if (candM != 0) push(candRef, candM);
            candRef = ref[k]; candM = m;
        }
        if (candM != 0) { curRef = candRef; curM = candM; have = true; }
    }
    if (team.diag) { team.tPacketEnd = __builtin_amdgcn_s_memrealtime(); team.entries = sp; }
    if (!dissolve) return occ;
    // lane per ray from here: every ray from the lowest node it is pending on, but not before a, and not beyond b
    uint32_t start = END;
    for (uint32_t e = 0; e < sp; ++e) {
        const uint64_t m = entryMask(e) & ~occ;
        if (m == 0) continue;
        const uint32_t eref = (uint32_t)__builtin_amdgcn_readlane((int)stRef, (int)e);
        const uint32_t self = *(ConstU32Ptr)(uintptr_t)(wideAddr + eref + 112u);
        if (__builtin_amdgcn_inverse_ballot_w64(m)) start = self < start ? self : start;
    }
    if (start != END && start < a) start = a;
    if (team.diag) team.tLaneStart = __builtin_amdgcn_s_memrealtime();
    const bool h = traverseShare<true, true, true>(bvh, r, start != END, start, lds, nullptr, p.parents, b, &team);
    return occ | __builtin_amdgcn_ballot_w64(h);
}

// One piece: the lanes (rays of the tile) it found occluded inside [a, b).
__device__ __forceinline__ uint64_t traversePiece(const TraceParams& p, const NodeStream& bvh, const Ray& r, bool live, uint32_t a, uint32_t b,
                                                  uint32_t entryRef, PieceShare& team, uint32_t* lds) {
    const uint64_t liveMask = __builtin_amdgcn_ballot_w64(live);
    if (liveMask == 0 || a >= b) return 0;                               // (an empty range: more pieces than the log had cuts for)
    const u32x8 root = *(ConstNodePtr)(uintptr_t)uniform64(p.bvh);
    const float rootLo[3] = { __uint_as_float(root.s0), __uint_as_float(root.s1), __uint_as_float(root.s2) };
    const float rootHi[3] = { __uint_as_float(root.s4), __uint_as_float(root.s5), __uint_as_float(root.s6) };
    WideRay w;
    const bool ok = wideRaySetup(r, rootLo, rootHi, w) && raySafe(r);
    if (__builtin_amdgcn_ballot_w64(live && !ok) != 0) {                 // the same decision in every piece of the tile
        if (a != 0u) return 0;
        return __builtin_amdgcn_ballot_w64(traverseShare<false>(bvh, r, live, 0u, lds));
    }
    return wideWalkRange(p, bvh, r, w, liveMask, a, b, entryRef, team, lds);
}

template <int VARIANT, bool FAST>
__device__ __forceinline__ bool traverse(const TraceParams& p, const NodeStream& bvh, const Ray& r, bool live, uint32_t* lds) {
    if (VARIANT == V_SHARE) return traverseShare<FAST>(bvh, r, live, 0u, lds);
    if (VARIANT == V_WHILEWHILE) return traverseWhileWhile<FAST>(bvh, r, live);
    if (VARIANT == V_POSTPONE) return traversePostpone<FAST>(bvh, r, live);
    return traverseStraight<FAST>(bvh, r, live);
}

// Tile row worked on by the k-th row of workgroups of a 2-D launch.  The hardware starts workgroups in grid order, so
// this is the order in which the rows of the image are begun -- and the rows begun last are the tail of the kernel, when
// the machine runs empty.  Order 2 (middle row first, then alternately below and above it) makes that tail the top and
// the bottom of the image: sky or ceiling, and the floor next to the camera -- on every scene measured the cheapest
// rows, while the rows that hold the long waves (tools/floor_analysis.py) start early.  A bijection on [0, blocksY):
// even k -> mid + k/2, odd k -> mid - (k+1)/2 with mid = blocksY/2.
__device__ __forceinline__ uint32_t dispatchRow(const TraceParams& p, uint32_t k) {
    if (p.rowOrder == 2u) {
        const uint32_t mid = p.blocksY >> 1;
        return (k & 1u) ? mid - ((k + 1u) >> 1) : mid + (k >> 1);
    }
    return p.rowOrder == 1u ? p.blocksY - 1u - k : k;
}

// ------------------------------------------------------------------------------------------------
// tile mapping: wave -> 8x8 pixel tile.  A 256-thread block is a 2x2 group of tiles (16x16 px).
// With swizzle on, blocks that share an XCD (blockIdx % 8, round-robin dispatch) get a contiguous
// chunk of the block sequence, and the sequence walks the image in 8-block-wide column strips so
// that co-resident blocks touch neighbouring subtrees of the BVH (per-XCD L2 locality).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool blockToXY(const TraceParams& p, uint32_t bid, uint32_t* bx, uint32_t* by) {
    if (p.grid2d) {                              // natural order, launched as a blocksX x blocksY grid: no division
        *bx = blockIdx.x;
        *by = dispatchRow(p, blockIdx.y);
        return true;
    }
    uint32_t b = bid;
    if (p.tileOrder) {                           // dispatch order given by the caller (longest tiles first)
        if (b >= p.nBlocks) return false;
        b = p.tileOrder[b];
        *by = b / p.blocksX;
        *bx = b - *by * p.blocksX;
        return true;
    }
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


// Row of the frame for the v-th row this dispatch owns.  Plain stripe: rowBegin + v.  Interleaved
// stripes (multi-GPU): bands of bandRows rows dealt round-robin, this device owns every
// nStripes-th band starting with band `stripe`.
__device__ __forceinline__ uint32_t ownedRow(const TraceParams& p, uint32_t v) {
    if (p.nStripes <= 1u) return p.rowBegin + v;
    const uint32_t band = v / p.bandRows;
    return (band * p.nStripes + p.stripe) * p.bandRows + (v - band * p.bandRows);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void shadowMaskKernel(TraceParams p) {
    __shared__ uint32_t shareSlots[4][64];       // lane numbers exchanged by traverseShare (256 B per wave)
    uint32_t* lds = shareSlots[threadIdx.x >> 6];
    uint32_t bx, by;
    if (!blockToXY(p, blockIdx.x, &bx, &by)) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = ownedRow(p, by * 16u + (wave >> 1) * 8u + (lane >> 3));
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
        Ray r = makeShadowRay(p, rel, s, (uint32_t)pix);
        bool unsafe = live && !raySafe(r);
        bool occluded;
        if (p.bvhFinite && __builtin_amdgcn_ballot_w64(unsafe) == 0)
            occluded = traverse<VARIANT, true>(p, bvh, r, live, lds);
        else
            occluded = traverse<VARIANT, false>(p, bvh, r, live, lds);
        lit += occluded ? 0u : 1u;                                  // comp:148
    }
    if (live) __builtin_nontemporal_store((uint8_t)lit, &p.mask[pix]);   // comp:150
}

// One piece of a split tile, start to end (called by the first pieceRows rows of a TILESPLIT launch): the tile's rays are
// set up as by its own wave (comp:128-146), walked over the piece's index range, and the piece that finishes last stores
// the tile.
template <bool BANDS>
__device__ __forceinline__ void runPiece(const TraceParams& p, uint32_t* lds) {
    const uint32_t pieceId = blockIdx.y * gridDim.x + blockIdx.x;
    if (pieceId >= p.nPieces) return;
    const u32x8 rec8 = *(ConstNodePtr)(uintptr_t)(uniform64(p.pieces) + (uint64_t)pieceId * 32u);
    const u32x4 rec{ rec8.s0, rec8.s1, rec8.s2, rec8.s3 };
    const uint32_t bx = rec.x & 0xFFFFu, by = rec.x >> 16, pieceCount = rec.w >> 24;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t x = bx * 8u + (lane & 7u);
    uint32_t y;
    if constexpr (BANDS) {
        const uint32_t band = by >> p.bandShift, within = by - (band << p.bandShift);
        y = (band * p.nStripes + p.stripe) * p.bandRows + within * 8u + (lane >> 3);
    } else y = p.rowBegin + by * 8u + (lane >> 3);
    const bool live = (x < p.W) && (y < p.rowEnd);
    const size_t pix = (size_t)y * p.W + x;
    F3 rel{ 0.f, 0.f, 0.f };
    if (live) {
        f32x4 t = __builtin_nontemporal_load((const f32x4*)p.positions + pix);          // comp:135
        rel = F3{ t.x, t.y, t.z };
    }
    const uint64_t tBegin = p.pieceClock ? __builtin_amdgcn_s_memrealtime() : 0;
    PieceShare team;
    team.diag = p.pieceClock != nullptr;
    team.state = p.tileState + (size_t)(rec.w & 0xFFFFFFu) * 2u;
    if (p.pieceLog) { team.log = p.pieceLog + (size_t)pieceId * (p.pieceLogCap + 1u); team.logCap = p.pieceLogCap; }
    const NodeStream bvh = openStream(p);
    const Ray r = makeShadowRay(p, rel, 0u, (uint32_t)pix);
    uint64_t tReady = 0;
    if (p.pieceClock) { asm volatile("" :: "v"(r.inv.x), "v"(r.inv.y), "v"(r.inv.z), "v"(r.o.x)); tReady = __builtin_amdgcn_s_memrealtime(); }
    const uint64_t occ = traversePiece(p, bvh, r, live, rec.y, rec.z, rec8.s4, team, lds);
    if (team.log && lane == 0) team.log[0] = team.logCount < team.logCap ? team.logCount : team.logCap;
    const uint64_t tWalked = p.pieceClock ? __builtin_amdgcn_s_memrealtime() : 0;
    // The piece that finishes last stores the tile.  Both atomics return a value, and the count is only added once the OR
    // has been performed (its result is an operand of the add as far as the compiler can tell), so the piece that reads
    // pieceCount - 1 finds every other piece's lanes in state[0].
    uint32_t done = 0;
    if (lane == 0) {
        const uint64_t before = __hip_atomic_fetch_or(team.state, occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t one = 1;
        asm volatile("; the count follows the OR" : "+v"(one) : "v"(before));
        done = (uint32_t)__hip_atomic_fetch_add(team.state + 1, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);
    if (p.pieceClock && lane == 0) {                                  // diagnostics: 8 u64 per piece
        uint64_t* o = p.pieceClock + (size_t)pieceId * 8u;
        o[0] = tBegin; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tReady; o[3] = team.tPacketEnd; o[4] = team.tLaneStart; o[5] = tWalked;
        o[6] = (uint64_t)team.nodes | ((uint64_t)team.entries << 32); o[7] = occ;
    }
    if (done + 1u != pieceCount) return;
    uint64_t all = 0;
    if (lane == 0) {
        all = __hip_atomic_fetch_or(team.state, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t zero = 0;
        asm volatile("; the reset follows the read" : "+v"(zero) : "v"(all));
        __hip_atomic_store(team.state, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(team.state + 1, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    all = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(all >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)all);
    if (live) __builtin_nontemporal_store((uint8_t)(((all >> lane) & 1ull) ? 0u : 1u), &p.mask[pix]);   // comp:148-150
}

// Packet kernels: a wave is a (8 or 16) x (8 or 16) pixel tile, a lane carries K = 1, 2 or 4 rays (the
// 8x8 sub-tiles of its wave tile), a 256-thread block is 2x2 wave tiles.
// WPB = waves per block: 4 (block = 2x2 wave tiles) or 1 (block = one wave tile, so that a finished wave
// frees its slot without waiting for three siblings).
// SOFT = more than one sample per pixel: only then the G-buffer position has to stay in registers across the walk.
// PLAIN = the everyday launch (natural tile order on a 2-D grid, one contiguous row range, no diagnostics): the scalar
// prologue that sorts out the other cases is compiled away.
// WIDE = the walk over the private wide nodes (K = 1, WPB = 1): 1 = the loop in assembly, 2 = the same loop compiled, 3 = 1
// with per-lane stacks in LDS for the lane-per-ray continuation over the wide nodes (option "wide_lane").
// SPLIT (soft shadows, K = 1, WPB = 1): SPLIT waves per workgroup work on the SAME 8x8 tile, wave w walks samples w,
// w + SPLIT, ...; the counts meet in LDS and wave 0 stores the byte.  A pixel's samples then run side by side instead of
// one after the other: a wave lives 1/SPLIT as long (shorter tail, finer-grained stripes).
// BANDS (with PLAIN): the everyday launch of ONE STRIPE of a frame cut into interleaved bands (multi-GPU):
// a band is 2^bandShift tile rows, so the frame row of a tile row is two shifts and a multiply on the scalar unit
// instead of the general prologue with its per-lane division.
// TILESPLIT (with PLAIN, one sample): the launch carries a split table (rts_ctx_plan_splits).  The first pieceRows rows of the
// grid are its records, dispatched first, so that the longest work starts first: PIECES of the tiles measured to be very long
// (traversePiece), then FRONT tiles -- long tiles that are not worth splitting, walked by their own wave as ever, only early.
// The remaining rows are the everyday tile waves, of which those of a tile of the table have nothing to do (one bitmap look-up).
template <int K, int WPB, bool PREFETCH = false, bool SOFT = false, bool PLAIN = false, int WIDE = 0, int SPLIT = 1, bool BANDS = false,
          bool TILESPLIT = false, bool PIECES = true>
// PIECES (with TILESPLIT): false = the table holds front tiles only (the whole-dispatch order of the 4K frames): the instantiation
// without the piece path -- the everyday path then keeps the registers the piece path's state would take.
// (Registers: a SIMD holds 8 waves of a kernel only up to 64 VGPRs AND 80 SGPRs including VCC / FLAT_SCRATCH / XNACK: the
//  next granule, 96, plus the 16 the trap handler adds per wave fits 800 only 7 times -- measured with the hardware slot ids
//  of the probe waves, DESIGN.md 4.7.  Every K = 1 instantiation is inside both limits; tools/gen_wide_asm.py budgets for it.)
__global__ __launch_bounds__(64 * WPB * SPLIT) __attribute__((amdgpu_waves_per_eu(K == 1 ? 8 : 4)))
void shadowMaskPacketKernel(TraceParams p) {
    static_assert(!BANDS || (PLAIN && K == 1 && WPB == 1), "the band form exists for the one-tile everyday launch only");
    static_assert(SPLIT == 1 || (K == 1 && WPB == 1 && SOFT), "samples are split over waves in the one-tile soft-shadow form only");
    static_assert(!TILESPLIT || (PLAIN && K == 1 && WPB == 1 && !SOFT && SPLIT == 1), "split tiles exist for the one-tile everyday launch only");
    __shared__ uint32_t shareSlots[WPB * SPLIT][64];     // lane numbers exchanged by traverseShare (256 B per wave)
    uint32_t* lds = shareSlots[threadIdx.x >> 6];
    // per-lane stacks of the wide lane walk: 4 KB per wave -- which caps a CU at 28 one-wave workgroups instead of 32, so
    // only the instantiations that use them (WIDE == 3: option "wide_lane") allocate them
    constexpr bool LANE_STACKS = WIDE == 3;
    __shared__ uint32_t laneStacks[LANE_STACKS ? WPB * SPLIT : 1][LANE_STACKS ? LANE_STACK * 64 : 1];
    uint32_t* laneStack = LANE_STACKS ? laneStacks[threadIdx.x >> 6] : nullptr;
    __shared__ uint32_t partial[SPLIT > 1 ? SPLIT : 1][SPLIT > 1 ? 64 : 1];                          // per-wave counts of unoccluded samples
    constexpr uint32_t TW = K >= 2 ? 16u : 8u, TH = K >= 4 ? 16u : 8u;
    uint32_t bx = blockIdx.x, by = 0;
    bool mine = true;
    if constexpr (PLAIN) {
        // Everything a tile wave needs before it can ask for its texel is the first 64 bytes of the argument block: asked for
        // here in one batch (the compiler would fetch each field where it is first used: three or four dependent round
        // trips to the scalar cache in front of the texel request).
        const uint64_t posAddr = (uint64_t)(uintptr_t)p.positions, mapAddr = (uint64_t)(uintptr_t)p.skipMap;
        const uint32_t a0 = p.W, a1 = p.rowBegin, a2 = p.rowEnd, a3 = p.pieceRows, a4 = p.blocksX, a5 = p.blocksY, a6 = p.rowOrder,
                       a7 = p.bandShift, a8 = p.stripe;
        asm volatile("" :: "s"(posAddr), "s"(mapAddr), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8));
    }
    if constexpr (TILESPLIT) {
        const uint32_t pieceRows = p.pieceRows, blocksX = p.blocksX, blocksY = p.blocksY, rowOrder = p.rowOrder;
        const uint64_t mapAddr = uniform64(p.skipMap);
        if (blockIdx.y < pieceRows) {                                 // the head of the grid: records of the split table
            const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
            if (id >= p.nPieces) return;
            const uint64_t mapFront = uniform64(p.frontMap);
            const uint32_t slot = (id & 7u) * p.frontStride + (id >> 3);       // (this XCD's run of the map: TraceParams::frontMap)
            const uint32_t tile = mapFront ? *(ConstU32Ptr)(uintptr_t)(mapFront + (uint64_t)slot * 4u) : 0xFFFFFFFFu;
            if (tile == 0xFFFFFFFFu) { if constexpr (PIECES) runPiece<BANDS>(p, lds); return; }   // a piece of a split tile: a path of its own
            bx = tile & 0xFFFFu; by = tile >> 16;                     // a FRONT tile: a long tile's own wave, started first
        } else {
            const uint32_t k = blockIdx.y - pieceRows;
            by = rowOrder == 1u ? blocksY - 1u - k : (rowOrder == 2u ? ((k & 1u) ? (blocksY >> 1) - ((k + 1u) >> 1) : (blocksY >> 1) + (k >> 1)) : k);
            // the tile's bit of the split table (the word travels with the next batch of kernel arguments)
            const uint32_t bit = by * blocksX + bx;
            const uint32_t word = *(ConstU32Ptr)(uintptr_t)(mapAddr + (uint64_t)(bit >> 5) * 4u);
            mine = !((word >> (bit & 31u)) & 1u);                    // a tile of the table is walked by its record(s) at the head
            if (!mine) return;                                       // (a scalar branch: nothing of this wave is needed)
        }
    } else by = dispatchRow(p, blockIdx.y);                           // (PLAIN: a 2-D grid, rows in dispatchRow order)
    if (!PLAIN && !blockToXY(p, blockIdx.x, &bx, &by)) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x0 = WPB == 4 ? bx * (2u * TW) + (wave & 1u) * TW + (lane & 7u) : bx * TW + (lane & 7u);     // (SPLIT: every wave, the same tile)
    const uint32_t v0 = (WPB == 4 ? by * (2u * TH) + (wave >> 1) * TH : by * TH) + (lane >> 3);
    bool live[K];
    size_t pix[K];
    F3 rel[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t x = x0 + (k & 1) * 8u;
        uint32_t y;
        if constexpr (BANDS) {
            const uint32_t band = by >> p.bandShift, within = by - (band << p.bandShift);
            y = (band * p.nStripes + p.stripe) * p.bandRows + within * 8u + (lane >> 3);
        } else y = PLAIN ? p.rowBegin + v0 + (k >> 1) * 8u : ownedRow(p, v0 + (k >> 1) * 8u);
        live[k] = (x < p.W) && (y < p.rowEnd) && mine;
        pix[k] = (size_t)y * p.W + x;
        if constexpr (PLAIN) {
            // (no branch around the request: a lane without a pixel asks for texel 0 and never looks at it -- with the branch
            //  the compiler waits for the texel inside it, before the rest of the prologue's scalar work)
            const f32x4 t = __builtin_nontemporal_load((const f32x4*)p.positions + (live[k] ? pix[k] : (size_t)0));   // comp:135
            rel[k] = F3{ t.x, t.y, t.z };
        } else {
            rel[k] = F3{ 0.f, 0.f, 0.f };
            if (live[k]) {
                f32x4 t = __builtin_nontemporal_load((const f32x4*)p.positions + pix[k]);   // comp:135
                rel[k] = F3{ t.x, t.y, t.z };
            }
        }
    }
    const NodeStream bvh = openStream(p);
    const uint32_t ns = SOFT ? p.nsamples : 1u;
    // clock probe (every instantiation, so that the clock is measured on the launches that are timed): one wave per tile row
    // (the stamps go straight to memory: nothing of the probe stays in registers across the walk)
    // (with a table: the tile rows stamp; when the table holds every tile there are none, and the front-tile rows stamp instead)
    const uint32_t probeRow = TILESPLIT ? (p.allInTable ? blockIdx.y : blockIdx.y - p.pieceRows) : (p.grid2d ? blockIdx.y : 0u);
    const bool probed = p.clockProbe != nullptr && blockIdx.x == 0 && threadIdx.x == 0 &&
                        (!TILESPLIT || (p.allInTable ? blockIdx.y < p.blocksY : blockIdx.y >= p.pieceRows));
    if (probed) {
        uint64_t* o = p.clockProbe + (size_t)probeRow * 4;
        o[0] = __builtin_amdgcn_s_memtime(); o[2] = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t tStart = !PLAIN && p.waveStats ? __builtin_amdgcn_s_memtime() : 0;   // diagnostics only
    const uint64_t rStart = !PLAIN && p.waveStats ? __builtin_amdgcn_s_memrealtime() : 0;
    int32_t left = 0;
    ShareDiag shareDiag;
    shareDiag.on = !PLAIN && p.waveStats != nullptr;
    uint32_t lit[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lit[k] = 0;
    uint64_t tReady = 0;
    for (uint32_t s = SPLIT > 1 ? wave : 0u; s < ns; s += SPLIT) {
        Ray r[K];
        bool occluded[K];
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = makeShadowRay(p, rel[k], s, (uint32_t)pix[k]);
        if (!PLAIN && p.waveStats && s < SPLIT) {    // diagnostics: the G-buffer texel is in and the first ray exists
            asm volatile("" :: "v"(r[0].inv.x), "v"(r[0].inv.y), "v"(r[0].inv.z), "v"(r[0].o.x));
            tReady = __builtin_amdgcn_s_memtime();
        }
        if constexpr (WIDE != 0) occluded[0] = traverseWide<WIDE != 2>(p, bvh, r[0], live[0], lds, laneStack, &left, &shareDiag);
        else traversePacket<K, PREFETCH>(p, bvh, r, live, occluded, lds, &left, &shareDiag);
#pragma unroll
        for (int k = 0; k < K; ++k) lit[k] += occluded[k] ? 0u : 1u;                     // comp:148
    }
    // (8-byte row stores built from a ballot were tried: WRITE_SIZE stayed at 40 MB per 8.3 MB mask -- the
    // memory side counts 32-byte sectors either way -- and the kernel got 10 % slower; byte stores stay.)
    if constexpr (SPLIT > 1) {
        partial[wave][lane] = lit[0];
        __syncthreads();
        if (wave == 0) {
            uint32_t sum = 0;
#pragma unroll
            for (int w = 0; w < SPLIT; ++w) sum += partial[w][lane];
            if (live[0]) __builtin_nontemporal_store((uint8_t)sum, &p.mask[pix[0]]);     // comp:150
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
#ifdef RTS_EXPERIMENT_NO_MASK_STORE      // (experiment build only, tools/mask_store_ab.sh: what the byte stores cost -- nothing is ever stored)
            if (live[k] && lit[k] > 200u) __builtin_nontemporal_store((uint8_t)lit[k], &p.mask[pix[k]]);
#else
            if (live[k]) __builtin_nontemporal_store((uint8_t)lit[k], &p.mask[pix[k]]);   // comp:150
#endif
    }
    if (probed) {
        uint64_t* o = p.clockProbe + (size_t)probeRow * 4;
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        // (the wave's hardware slot rides in the top 16 bits of the end stamp: 2^48 shader clocks are 32 hours)
        o[1] = (__builtin_amdgcn_s_memtime() & 0x0000FFFFFFFFFFFFull) | ((uint64_t)(hwid & 0xFFFFu) << 48);
        o[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (!PLAIN && p.waveStats && lane == 0) {    // diagnostics: never read by any kernel, never part of an output
        const size_t slot = (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (WPB * SPLIT) + wave;
        uint64_t* o = p.waveStats + slot * 4;
        o[0] = tStart;
        o[1] = __builtin_amdgcn_s_memtime();
        // shader clocks against the 100 MHz reference over the same interval: the clock the chip held under this load
        p.waveRealtime[slot * 4] = rStart;
        p.waveRealtime[slot * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        p.waveRealtime[slot * 4 + 2] = tReady - tStart;      // clocks from wave start to "first ray ready"
        uint32_t xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));       // wave slot, SIMD, CU, SE ... of this wave
        p.waveRealtime[slot * 4 + 3] = ((uint64_t)hwid << 32) | xcc;
        // dissolved flag | lane-per-ray iterations after the dissolve | clocks from start to the dissolve
        o[2] = (left < 0 ? 1ull : 0ull) | ((uint64_t)(shareDiag.iterations & 0xFFFFFFu) << 8) |
               ((shareDiag.tDissolve ? (shareDiag.tDissolve - tStart) & 0xFFFFFFFFull : 0ull) << 32);
        o[3] = ((uint64_t)bx << 48) | ((uint64_t)(by & 0xFFFFu) << 32) | shareDiag.laneSteps;   // ... and the lane-steps in them
    }
}

template <int VARIANT>
__global__ __launch_bounds__(256) void traceRaysKernel(TraceParams p) {
    __shared__ uint32_t shareSlots[4][64];
    uint32_t* lds = shareSlots[threadIdx.x >> 6];
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
    bool occluded;
    if (VARIANT >= V_PACKET && VARIANT != V_SHARE) {   // 64 consecutive rays as one packet (falls apart if incoherent)
        const Ray rr[1] = { r };
        const bool ll[1] = { live };
        bool oo[1];
        traversePacket<1>(p, bvh, rr, ll, oo, lds);
        occluded = oo[0];
    } else {
        bool unsafe = live && !raySafe(r);
        if (p.bvhFinite && __builtin_amdgcn_ballot_w64(unsafe) == 0)
            occluded = traverse<VARIANT, true>(p, bvh, r, live, lds);
        else
            occluded = traverse<VARIANT, false>(p, bvh, r, live, lds);
    }
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
    case V_PACKET: return mask ? "shadowMaskPacketKernel<1>" : "traceRaysKernel<3>";
    case V_PACKET2: return mask ? "shadowMaskPacketKernel<2>" : "traceRaysKernel<3>";
    case V_PACKET4: return mask ? "shadowMaskPacketKernel<4>" : "traceRaysKernel<3>";
    case V_PACKET_PF: return mask ? "shadowMaskPacketKernel<1,pf>" : "traceRaysKernel<3>";
    case V_SHARE: return mask ? "shadowMaskKernel<7>" : "traceRaysKernel<7>";
    case V_WIDE: return mask ? "shadowMaskPacketKernel<1,wide>" : "traceRaysKernel<7>";
    case V_WIDE_C: return mask ? "shadowMaskPacketKernel<1,wide,compiled>" : "traceRaysKernel<7>";
    }
    return "?";
}

void tileShape(int variant, int wavesPerBlock, uint32_t* blockW, uint32_t* blockH) {
    const bool packet = variant >= V_PACKET && variant <= V_PACKET_PF;
    if (variant == V_WIDE_C) { *blockW = 8u; *blockH = 8u; return; }           // always one wave tile per workgroup
    if (variant == V_WIDE) { *blockW = *blockH = wavesPerBlock == 4 ? 16u : 8u; return; }
    const uint32_t f = (packet && wavesPerBlock == 1) ? 1u : 2u;                // block = f x f wave tiles
    *blockW = f * ((variant == V_PACKET2 || variant == V_PACKET4) ? 16u : 8u);
    *blockH = f * (variant == V_PACKET4 ? 16u : 8u);
}

hipError_t launchShadowMask(int variant, int wavesPerBlock, const TraceParams& p, hipStream_t stream, uint32_t ldsPad) {
    dim3 grid(p.gridBlocks), block(256);
    // (the records of a split table come first; when every tile has a record there are no tile rows at all)
    if (p.grid2d) grid = dim3(p.blocksX, (p.pieces && p.allInTable ? 0u : p.blocksY) + (p.pieces ? p.pieceRows : 0u));
    const bool soft = p.nsamples > 1;
    if (variant == V_WIDE && wavesPerBlock == 4) {                       // 2 x 2 tiles per workgroup: neighbours share the scalar cache
        if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, false, true, false, 1>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, false, false, false, 1>), grid, block, 0, stream, p);
        return hipGetLastError();
    }
    if (variant == V_WIDE) {
        dim3 b1(64);
        if (soft && p.softSplit) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, true, false, 1, 4>), grid, dim3(256), 0, stream, p);
        else if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, true, false, 1>), grid, b1, ldsPad, stream, p);
        else if (p.wideLane)                        // lane-per-ray continuation over the wide nodes: the instantiation with LDS stacks
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, false, 3>), grid, b1, ldsPad, stream, p);
        else if (p.pieces && p.nStripes > 1 && p.hasPieces)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1, 1, true, true>), grid, b1, ldsPad, stream, p);
        else if (p.pieces && p.nStripes > 1)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1, 1, true, true, false>), grid, b1, ldsPad, stream, p);
        else if (p.pieces && p.hasPieces)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1, 1, false, true>), grid, b1, ldsPad, stream, p);
        else if (p.pieces)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1, 1, false, true, false>), grid, b1, ldsPad, stream, p);
        else if (p.grid2d && p.nStripes > 1 && p.bandShift != 0xFFFFFFFFu && !p.waveStats && p.rowOrder == 0)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1, 1, true>), grid, b1, ldsPad, stream, p);
        else if (p.grid2d && p.nStripes <= 1 && !p.waveStats)
            hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 1>), grid, b1, ldsPad, stream, p);
        else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, false, 1>), grid, b1, ldsPad, stream, p);
        return hipGetLastError();
    }
    if (variant == V_WIDE_C) {
        dim3 b1(64);
        if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, true, false, 2>), grid, b1, ldsPad, stream, p);
        else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, false, 2>), grid, b1, ldsPad, stream, p);
        return hipGetLastError();
    }
    if (variant >= V_PACKET && variant <= V_PACKET_PF && wavesPerBlock == 1) {
        dim3 b1(64);
        switch (variant) {
        case V_PACKET:
            if (soft && p.softSplit) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, true, false, 0, 4>), grid, dim3(256), 0, stream, p);
            else if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, true>), grid, b1, ldsPad, stream, p);
            else if (p.pieces && p.nStripes > 1 && p.hasPieces)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 0, 1, true, true>), grid, b1, ldsPad, stream, p);
            else if (p.pieces && p.nStripes > 1)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 0, 1, true, true, false>), grid, b1, ldsPad, stream, p);
            else if (p.pieces && p.hasPieces)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 0, 1, false, true>), grid, b1, ldsPad, stream, p);
            else if (p.pieces)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 0, 1, false, true, false>), grid, b1, ldsPad, stream, p);
            else if (p.grid2d && p.nStripes > 1 && p.bandShift != 0xFFFFFFFFu && !p.waveStats && p.rowOrder == 0)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true, 0, 1, true>), grid, b1, ldsPad, stream, p);
            else if (p.grid2d && p.nStripes <= 1 && !p.waveStats)
                hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false, true>), grid, b1, ldsPad, stream, p);
            else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, false, false>), grid, b1, ldsPad, stream, p);
            break;
        case V_PACKET2: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<2, 1, false, true>), grid, b1, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<2, 1, false, false>), grid, b1, 0, stream, p); break;
        case V_PACKET4: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<4, 1, false, true>), grid, b1, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<4, 1, false, false>), grid, b1, 0, stream, p); break;
        case V_PACKET_PF: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, true, true>), grid, b1, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 1, true, false>), grid, b1, 0, stream, p); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (variant) {
    case V_STRAIGHT: hipLaunchKernelGGL(shadowMaskKernel<V_STRAIGHT>, grid, block, 0, stream, p); break;
    case V_WHILEWHILE: hipLaunchKernelGGL(shadowMaskKernel<V_WHILEWHILE>, grid, block, 0, stream, p); break;
    case V_POSTPONE: hipLaunchKernelGGL(shadowMaskKernel<V_POSTPONE>, grid, block, 0, stream, p); break;
    case V_SHARE: hipLaunchKernelGGL(shadowMaskKernel<V_SHARE>, grid, block, 0, stream, p); break;
    case V_PACKET: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, false, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, false, false>), grid, block, 0, stream, p); break;
    case V_PACKET2: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<2, 4, false, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<2, 4, false, false>), grid, block, 0, stream, p); break;
    case V_PACKET4: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<4, 4, false, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<4, 4, false, false>), grid, block, 0, stream, p); break;
    case V_PACKET_PF: if (soft) hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, true, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((shadowMaskPacketKernel<1, 4, true, false>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// split planning: visit logs -> piece table.  One workgroup per selected tile: the S - 1 cuts of its index range are the
// j/S quantiles of the node indices its planning walk logged (a histogram of 1024 bins over the value range, refined inside
// the bin that holds the quantile until a bin is one index wide).  Piece j of the tile is [cut_j, cut_j+1), cut_0 = 0,
// cut_S = END; a tile whose log is empty gets one piece that covers everything and S - 1 empty ones.
// ------------------------------------------------------------------------------------------------
// The wide node a piece starts at: the lowest one whose subtree holds all of [a, b) -- found by walking down from the root
// while one inner slot's index range [lo_k, idx_k+1) holds the piece's (the levels skipped only cull: the piece's hits are
// confirmed against the leaf's parent box like every hit of the wide kernels).
__device__ uint32_t pieceEntry(const uint32_t* wide, uint32_t a, uint32_t b) {
    uint32_t ref = 0, end = END;
    for (int depth = 0; depth < 512; ++depth) {
        const uint32_t* n = wide + (ref >> 2);
        const uint32_t self = n[28];
        uint32_t next = END, nextEnd = END;
        for (int k = 0; k < 4; ++k) {
            const uint32_t rk = n[24 + k];
            if (rk == END || (rk & 1u)) continue;
            const uint32_t lo = k ? n[28 + k] : self + 1u;
            const uint32_t hi = (k < 3 && n[24 + k + 1] != END) ? n[28 + k + 1] : end;
            if (lo <= a && b <= hi) { next = rk; nextEnd = hi; }
        }
        if (next == END) break;
        ref = next; end = nextEnd;
    }
    return ref;
}

__global__ __launch_bounds__(256) void splitQuantilesKernel(const uint32_t* log, uint32_t cap, const SplitCut* cuts, const uint32_t* firstPiece,
                                                            uint32_t tiles, uint32_t* pieces, const uint32_t* wide) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t box[4];                       // [0] min, [1] max, then the refinement's {lo, hi}
    __shared__ uint32_t below;
    const uint32_t t = blockIdx.x;
    if (t >= tiles) return;
    const uint32_t* L = log + (size_t)t * (cap + 1u);
    const uint32_t n = L[0] < cap ? L[0] : cap, S = cuts[t].pieces, base = firstPiece[t];
    if (threadIdx.x == 0) { box[0] = END; box[1] = 0; }
    __syncthreads();
    uint32_t mn = END, mx = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) { const uint32_t v = L[1u + i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    atomicMin(&box[0], mn); atomicMax(&box[1], mx);
    __syncthreads();
    mn = box[0]; mx = box[1];
    uint32_t prevCut = 0;
    for (uint32_t j = 0; j < S; ++j) {
        uint32_t cut = END;                                             // the end of piece j
        if (j + 1 < S && n != 0) {
            const uint32_t target = (uint32_t)(((uint64_t)(j + 1) * n) / S);
            __syncthreads();
            if (threadIdx.x == 0) { box[2] = mn; box[3] = mx; below = 0; }
            for (;;) {
                __syncthreads();
                const uint32_t lo = box[2], hi = box[3];
                const uint32_t width = (uint32_t)(((uint64_t)(hi - lo) + 1024ull) / 1024ull);      // >= 1
                for (uint32_t i = threadIdx.x; i < 1024; i += 256) hist[i] = 0;
                __syncthreads();
                for (uint32_t i = threadIdx.x; i < n; i += 256) {
                    const uint32_t v = L[1u + i];
                    if (v >= lo && v <= hi) atomicAdd(&hist[(v - lo) / width], 1u);
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t cum = below, bin = 0;
                    while (bin < 1023 && cum + hist[bin] <= target) { cum += hist[bin]; ++bin; }
                    below = cum;
                    box[2] = lo + bin * width;
                    const uint64_t top = (uint64_t)lo + (uint64_t)(bin + 1) * width - 1ull;
                    box[3] = top < hi ? (uint32_t)top : hi;
                }
                __syncthreads();
                if (width == 1) break;
            }
            cut = box[2];
        }
        if (cut < prevCut) cut = prevCut;
        if (threadIdx.x == 0) {
            uint32_t* o = pieces + (size_t)(base + j) * 8u;
            o[0] = cuts[t].tile; o[1] = prevCut; o[2] = cut; o[3] = t | (S << 24);
            o[4] = prevCut < cut ? pieceEntry(wide, prevCut, cut) : 0u; o[5] = 0; o[6] = 0; o[7] = 0;
        }
        prevCut = cut;
    }
}

hipError_t launchSplitQuantiles(const uint32_t* d_log, uint32_t logCap, const SplitCut* d_cuts, const uint32_t* d_firstPiece,
                                uint32_t tiles, uint32_t* d_pieces, const void* d_wide, hipStream_t stream) {
    if (tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(splitQuantilesKernel, dim3(tiles), dim3(256), 0, stream, d_log, logCap, d_cuts, d_firstPiece, tiles, d_pieces,
                       (const uint32_t*)d_wide);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// self-test of rcpFast (the exactness argument of the fast reciprocal is this run): every bit pattern x with rcpInRange(x)
// must give the bits of 1.0f / x; counts[0] = patterns in range, counts[1] = of those, patterns that differ, counts[2] = an
// example.  2^32 divisions: a fraction of a second.
// ------------------------------------------------------------------------------------------------
__global__ void reciprocalSelfTestKernel(unsigned long long* counts) {
    unsigned long long in = 0, bad = 0;
    uint32_t example = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        if (!rcpInRange(x)) continue;
        ++in;
        if (__float_as_uint(rcpFast(x)) != __float_as_uint(1.0f / x)) { ++bad; example = (uint32_t)i; }
    }
    atomicAdd(&counts[0], in);
    if (bad) { atomicAdd(&counts[1], bad); counts[2] = example; }
}

hipError_t launchReciprocalSelfTest(unsigned long long* d_counts, hipStream_t stream) {
    hipLaunchKernelGGL(reciprocalSelfTestKernel, dim3(256 * 16), dim3(256), 0, stream, d_counts);
    return hipGetLastError();
}

hipError_t launchTraceRays(int variant, const TraceParams& p, hipStream_t stream) {
    dim3 grid((unsigned)((p.nrays + 255) / 256)), block(256);
    switch (variant) {
    case V_STRAIGHT: hipLaunchKernelGGL(traceRaysKernel<V_STRAIGHT>, grid, block, 0, stream, p); break;
    case V_WHILEWHILE: hipLaunchKernelGGL(traceRaysKernel<V_WHILEWHILE>, grid, block, 0, stream, p); break;
    case V_POSTPONE: hipLaunchKernelGGL(traceRaysKernel<V_POSTPONE>, grid, block, 0, stream, p); break;
    case V_SHARE: case V_WIDE: case V_WIDE_C: hipLaunchKernelGGL(traceRaysKernel<V_SHARE>, grid, block, 0, stream, p); break;
    case V_PACKET: case V_PACKET2: case V_PACKET4: case V_PACKET_PF:
        hipLaunchKernelGGL(traceRaysKernel<V_PACKET>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace rts
