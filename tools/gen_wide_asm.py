#!/usr/bin/env python3
"""Generates raytracedshadows_amd/csrc/rts_wide_asm.inc: the hand-written gfx950 loop of the WIDE packet kernel
(rts_kernels.hip, 'V_WIDE'; data layout: rts_wide.hip), instantiated for the 8 sign octants of the ordered slab test and
the generic form.  The .inc file is committed; re-run after editing:

    python tools/gen_wide_asm.py

One iteration = one wide node: two s_load_dwordx16 (128 B), four CHEAP slab tests (10 VALU each, EXEC = the members, so
the compare results are membership masks as they come), then per slot that somebody hit: a leaf slot runs the triangle
test for the lanes that hit its box and confirms triangle hits with the EXACT slab test of that box (comp:61-73); the
first inner slot becomes the next node, further ones are pushed (node, mask) on the stack that lives in three VGPRs
(entry i in lane i; v_writelane / v_readlane through M0).

Fixed scratch SGPRs (declared as clobbers), relative to SGPR_BASE = 20 -- 54 of them, so that the kernel stays within
80 SGPRs: the granule above (96) plus the 16 the trap handler adds per wave leaves room for 7 waves per SIMD, not 8
(measured with HW_ID stamps):
    N[0:27]   +0..+27   the wide node: slot k = N[6k..6k+2] bboxMin, N[6k+3..6k+5] bboxMax; N[24+k] ref  (dwords 28..31 of
                        the node are not loaded: x16 + x8 + x4)
    T[0:8]    +28..+36  triangle record: v0.xyz, e0.xyz, e1.xyz  (x8 + one dword; the rest of the record is for the lane walks)
    H0..H3    +38..+45  members that hit slot k          M   +46,47  members of the current node
    NXM       +48,49    members of the next node        NXREF +50   its ref (-1: none yet)
    REF       +51       byte offset of the current node  R   +52,53  reject mask / temporaries (the second temporary mask is VCC)
EXEC at entry is kept in lane 63 of the stack's two mask registers.  A node is only taken up with at most 59 entries on the
stack: it pushes at most 3 (62), and the dissolve that follows writes the current node back as entry 62 at the latest, so
lane 63 is never written by the stack (tests/test_gpu_parity.py: test_wide_stack_limit_dissolves_and_keeps_every_pixel).
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "raytracedshadows_amd", "csrc", "rts_wide_asm.inc")
BASE = 20
PREFETCH = False    # pushed nodes requested at push time: measured, no effect
AX = "xyz"


def s(i):
    return f"s{BASE + i}"


def sp(i):
    return f"s[{BASE + i}:{BASE + i + 1}]"


def N(i):
    return s(i)


def T(i):
    return s(28 + i)


HBASE = 38


def H(k):
    return sp(HBASE + 2 * k)


M, NXM, NXREF, REF, R, R2 = sp(46), sp(48), s(50), s(51), sp(52), "vcc"
MLO, MHI = s(46), s(47)
RLO, RHI = s(52), s(53)
NSGPR = 54


def lo(k, a):
    return N(6 * k + a)


def hi(k, a):
    return N(6 * k + 3 + a)


def far_near(k, a, octant):
    neg = (octant >> a) & 1
    return (lo(k, a), hi(k, a)) if neg else (hi(k, a), lo(k, a))


def cheap_pair(ka, kb, octant):
    """cheap slab tests of two slots, interleaved (12 temporaries): 10 VALU each."""
    L = []
    if octant < 8:
        for j, k in enumerate((ka, kb)):
            b = 6 * j
            for i, a in enumerate(AX):
                L.append(f"v_fma_f32 %[t{b + i}], {far_near(k, i, octant)[0]}, %[i{a}], -%[cu{a}]")
            for i, a in enumerate(AX):
                L.append(f"v_fma_f32 %[t{b + 3 + i}], {far_near(k, i, octant)[1]}, %[i{a}], -%[cd{a}]")
        for j, k in enumerate((ka, kb)):
            b = 6 * j
            L += [f"v_min3_f32 %[t{b}], %[t{b}], %[t{b + 1}], %[t{b + 2}]",
                  f"v_max_f32 %[t{b + 3}], %[t{b + 3}], %[t{b + 4}]",
                  f"v_max3_f32 %[t{b + 3}], %[t{b + 3}], %[t{b + 5}], 0"]
        for j, k in enumerate((ka, kb)):
            b = 6 * j
            L.append(f"v_cmp_ge_f32 {H(k)}, %[t{b}], %[t{b + 3}]")
    else:
        # generic: both planes against both constants; upper bounds with cu, lower bounds with cd (16 VALU per slot)
        for k in (ka, kb):
            for i, a in enumerate(AX):
                L += [f"v_fma_f32 %[t{i}], {hi(k, i)}, %[i{a}], -%[cu{a}]",
                      f"v_fma_f32 %[t{3 + i}], {lo(k, i)}, %[i{a}], -%[cu{a}]",
                      f"v_max_f32 %[t{i}], %[t{i}], %[t{3 + i}]",
                      f"v_fma_f32 %[t{6 + i}], {hi(k, i)}, %[i{a}], -%[cd{a}]",
                      f"v_fma_f32 %[t{9 + i}], {lo(k, i)}, %[i{a}], -%[cd{a}]",
                      f"v_min_f32 %[t{6 + i}], %[t{6 + i}], %[t{9 + i}]"]
            L += ["v_min3_f32 %[t0], %[t0], %[t1], %[t2]",
                  "v_max_f32 %[t6], %[t6], %[t7]",
                  "v_max3_f32 %[t6], %[t6], %[t8], 0",
                  f"v_cmp_ge_f32 {H(k)}, %[t0], %[t6]"]
    return L


def exact_box(k, octant, dst):
    """the reference's slab test (comp:61-73) of slot k's box for the lanes in EXEC -> dst (SGPR pair)."""
    L = []
    if octant < 8:
        for i, a in enumerate(AX):
            L.append(f"v_sub_f32 %[t{i}], {far_near(k, i, octant)[0]}, %[o{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_sub_f32 %[t{3 + i}], {far_near(k, i, octant)[1]}, %[o{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_mul_f32 %[t{i}], %[t{i}], %[i{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_mul_f32 %[t{3 + i}], %[t{3 + i}], %[i{a}]")
        L += ["v_min3_f32 %[t0], %[t0], %[t1], %[t2]",
              "v_max_f32 %[t3], %[t3], %[t4]",
              "v_max3_f32 %[t3], %[t3], %[t5], 0",
              f"v_cmp_ge_f32 {dst}, %[t0], %[t3]"]
    else:
        for i, a in enumerate(AX):
            L.append(f"v_sub_f32 %[t{i}], {hi(k, i)}, %[o{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_sub_f32 %[t{3 + i}], {lo(k, i)}, %[o{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_mul_f32 %[t{i}], %[t{i}], %[i{a}]")
        for i, a in enumerate(AX):
            L.append(f"v_mul_f32 %[t{3 + i}], %[t{3 + i}], %[i{a}]")
        L += ["v_max_f32 %[t6], %[t0], %[t3]", "v_min_f32 %[t0], %[t0], %[t3]",
              "v_max_f32 %[t3], %[t1], %[t4]", "v_min_f32 %[t1], %[t1], %[t4]",
              "v_max_f32 %[t4], %[t2], %[t5]", "v_min_f32 %[t2], %[t2], %[t5]",
              "v_min3_f32 %[t6], %[t6], %[t3], %[t4]",
              "v_max_f32 %[t0], %[t0], %[t1]",
              "v_max3_f32 %[t0], %[t0], %[t2], 0",
              f"v_cmp_ge_f32 {dst}, %[t6], %[t0]"]
    return L


def triangle(dst, tag=0):
    """Moeller-Trumbore exactly as comp:41-59 spells it (separate mul/sub/add, dots left to right, correctly rounded
    1/det: the sequence hipcc emits for 1.0f / x); v0 = T0-2, e0 = T3-5, e1 = T6-8.  dst = lanes of EXEC that are REJECTED.
    Ten temporaries: s2 takes the registers of s1 once b1 exists, b2 / t / b1+b2 those of dd once s2 exists (the number of
    VGPRs named by the statement decides whether a SIMD holds 8 waves of this kernel or 7)."""
    e0, e1, v0 = (T(3), T(4), T(5)), (T(6), T(7), T(8)), (T(0), T(1), T(2))
    return [
        f"v_mul_f32 %[t0], {e1[2]}, %[dy]", f"v_mul_f32 %[t1], {e1[1]}, %[dz]", "v_sub_f32 %[t0], %[t0], %[t1]",   # s1.x = d.y*e1.z - e1.y*d.z
        f"v_mul_f32 %[t1], {e1[0]}, %[dz]", f"v_mul_f32 %[t2], {e1[2]}, %[dx]", "v_sub_f32 %[t1], %[t1], %[t2]",   # s1.y = d.z*e1.x - e1.z*d.x
        f"v_mul_f32 %[t2], {e1[1]}, %[dx]", f"v_mul_f32 %[t3], {e1[0]}, %[dy]", "v_sub_f32 %[t2], %[t2], %[t3]",   # s1.z = d.x*e1.y - e1.x*d.y
        f"v_mul_f32 %[t3], {e0[0]}, %[t0]", f"v_mul_f32 %[t4], {e0[1]}, %[t1]", "v_add_f32 %[t3], %[t3], %[t4]",
        f"v_mul_f32 %[t4], {e0[2]}, %[t2]", "v_add_f32 %[t3], %[t3], %[t4]",                                       # det = dot(s1, e0)
        # 1/det: rcp + one Newton step where every lane's |det| is in [2^-100, 2^100) (the IEEE quotient there: every bit pattern
        # checked on the device, rts_selftest_reciprocal); otherwise the general division
        "v_bfe_u32 %[t5], %[t3], 23, 8",
        "v_subrev_u32 %[t5], 27, %[t5]",
        "v_cmp_lt_u32 vcc, 0xc7, %[t5]",
        f"s_cbranch_vccnz 17{tag}f",
        "v_rcp_f32 %[t4], %[t3]",
        "s_nop 0",                                       # (gfx940+: a VALU op that reads a transcendental's result needs one wait state; the assembler adds none)
        "v_fma_f32 %[t5], -%[t3], %[t4], 1.0",
        "v_fma_f32 %[t4], %[t5], %[t4], %[t4]",
        f"s_branch 18{tag}f",
        f"17{tag}:",
        f"v_div_scale_f32 %[t5], {R}, %[t3], %[t3], 1.0",
        "v_rcp_f32 %[t7], %[t5]",
        "v_div_scale_f32 %[t6], vcc, 1.0, %[t3], 1.0",
        "v_fma_f32 %[t8], -%[t5], %[t7], 1.0",
        "v_fmac_f32 %[t7], %[t8], %[t7]",
        "v_mul_f32 %[t8], %[t6], %[t7]",
        "v_fma_f32 %[t9], -%[t5], %[t8], %[t6]",
        "v_fmac_f32 %[t8], %[t9], %[t7]",
        "v_fma_f32 %[t5], -%[t5], %[t8], %[t6]",
        "v_div_fmas_f32 %[t5], %[t5], %[t7], %[t8]",
        "v_div_fixup_f32 %[t4], %[t5], %[t3], 1.0",                                                                # invd
        f"18{tag}:",
        f"v_subrev_f32 %[t5], {v0[0]}, %[ox]", f"v_subrev_f32 %[t6], {v0[1]}, %[oy]", f"v_subrev_f32 %[t7], {v0[2]}, %[oz]",   # dd = o - v0
        "v_mul_f32 %[t8], %[t5], %[t0]", "v_mul_f32 %[t9], %[t6], %[t1]", "v_add_f32 %[t8], %[t8], %[t9]",
        "v_mul_f32 %[t9], %[t7], %[t2]", "v_add_f32 %[t8], %[t8], %[t9]", "v_mul_f32 %[t8], %[t8], %[t4]",       # b1 = dot(dd, s1) * invd
        # reject = b1<0 || b1>1 || b2<0 || b1+b2>1 || t<0 || t>tmax   (ordered compares: false on NaN, comp:51).  A disjunction: once
        # EVERY lane of EXEC is rejected by the conditions evaluated so far, the rest of the test cannot change anything and is skipped
        f"v_cmp_gt_f32 {dst}, 0, %[t8]",
        f"v_cmp_lt_f32 {R2}, 1.0, %[t8]", f"s_or_b64 {dst}, {dst}, {R2}",
        f"s_andn2_b64 {R2}, exec, {dst}",
        f"s_cbranch_scc0 19{tag}f",
        f"v_mul_f32 %[t0], {e0[2]}, %[t6]", f"v_mul_f32 %[t3], {e0[1]}, %[t7]", "v_sub_f32 %[t0], %[t0], %[t3]",       # s2.x = dd.y*e0.z - e0.y*dd.z
        f"v_mul_f32 %[t1], {e0[0]}, %[t7]", f"v_mul_f32 %[t3], {e0[2]}, %[t5]", "v_sub_f32 %[t1], %[t1], %[t3]",       # s2.y = dd.z*e0.x - e0.z*dd.x
        f"v_mul_f32 %[t2], {e0[1]}, %[t5]", f"v_mul_f32 %[t3], {e0[0]}, %[t6]", "v_sub_f32 %[t2], %[t2], %[t3]",       # s2.z = dd.x*e0.y - e0.x*dd.y
        "v_mul_f32 %[t5], %[dx], %[t0]", "v_mul_f32 %[t6], %[dy], %[t1]", "v_add_f32 %[t5], %[t5], %[t6]",
        "v_mul_f32 %[t6], %[dz], %[t2]", "v_add_f32 %[t5], %[t5], %[t6]", "v_mul_f32 %[t5], %[t5], %[t4]",         # b2 = dot(d, s2) * invd
        "v_add_f32 %[t7], %[t8], %[t5]",                                                                             # b1 + b2
        f"v_cmp_gt_f32 {R2}, 0, %[t5]", f"s_or_b64 {dst}, {dst}, {R2}",
        f"v_cmp_lt_f32 {R2}, 1.0, %[t7]", f"s_or_b64 {dst}, {dst}, {R2}",
        f"s_andn2_b64 {R2}, exec, {dst}",
        f"s_cbranch_scc0 19{tag}f",
        f"v_mul_f32 %[t6], {e1[0]}, %[t0]", f"v_mul_f32 %[t7], {e1[1]}, %[t1]", "v_add_f32 %[t6], %[t6], %[t7]",
        f"v_mul_f32 %[t7], {e1[2]}, %[t2]", "v_add_f32 %[t6], %[t6], %[t7]", "v_mul_f32 %[t6], %[t6], %[t4]",       # t = dot(e1, s2) * invd
        f"v_cmp_gt_f32 {R2}, 0, %[t6]", f"s_or_b64 {dst}, {dst}, {R2}",
        f"v_cmp_lt_f32 {R2}, %[tm], %[t6]", f"s_or_b64 {dst}, {dst}, {R2}",          # (tmax is the same for every ray of a launch: an SGPR)
        f"19{tag}:"]


def loop(octant):
    L = [f"s_mov_b64 {R}, exec",
         f"v_writelane_b32 %[vlo], {RLO}, 63",
         f"v_writelane_b32 %[vhi], {RHI}, 63",
         f"s_mov_b32 {REF}, 0",                                  # the root's wide node
         f"s_andn2_b64 {M}, %[live], %[occ]",
         "s_cbranch_scc0 90f",
         # ---- one node --------------------------------------------------------------------------------------------
         "1:",
         f"s_load_dwordx16 s[{BASE}:{BASE + 15}], %[wb], {REF}",
         f"s_load_dwordx8 s[{BASE + 16}:{BASE + 23}], %[wb], {REF} offset:64",
         f"s_load_dwordx4 s[{BASE + 24}:{BASE + 27}], %[wb], {REF} offset:96",
         "s_cmp_gt_u32 %[sp], 59",                               # a node pushes at most 3 entries; entry 63 is the saved EXEC
         "s_cbranch_scc1 80f",
         f"s_mov_b32 {NXREF}, -1",
         f"s_mov_b64 exec, {M}",
         "s_waitcnt lgkmcnt(0)"]
    L += cheap_pair(0, 1, octant)
    L += cheap_pair(2, 3, octant)
    for k in range(4):
        L += [f"s_cmp_lg_u64 {H(k)}, 0",
              f"s_cbranch_scc0 2{k}f",
              f"s_bitcmp1_b32 {N(24 + k)}, 0",
              f"s_cbranch_scc1 3{k}f",                          # leaf slot (out of line)
              f"s_cmp_eq_u32 {NXREF}, -1",
              f"s_cbranch_scc0 4{k}f",                          # a next node is chosen already: push (out of line)
              f"s_mov_b32 {NXREF}, {N(24 + k)}",
              f"s_mov_b64 {NXM}, {H(k)}",
              f"2{k}:"]
    L += [f"s_cmp_eq_u32 {NXREF}, -1",
          "s_cbranch_scc1 5f",
          f"s_mov_b32 {REF}, {NXREF}",
          f"s_andn2_b64 {M}, {NXM}, %[occ]",
          "s_cbranch_scc1 1b",
          # ---- pop -----------------------------------------------------------------------------------------------
          "5:",
          "s_cmp_eq_u32 %[sp], 0",
          "s_cbranch_scc1 90f",
          "s_sub_u32 %[sp], %[sp], 1",
          "s_mov_b32 m0, %[sp]",
          f"v_readlane_b32 {REF}, %[vref], m0",
          f"v_readlane_b32 {MLO}, %[vlo], m0",
          f"v_readlane_b32 {MHI}, %[vhi], m0",
          f"s_andn2_b64 {M}, {M}, %[occ]",
          "s_cbranch_scc0 5b",
          # coherence statistics at pops: members picked up against rays alive (dissolve rule)
          f"s_bcnt1_i32_b64 {RLO}, {M}",
          f"s_add_u32 %[acc], %[acc], {RLO}",
          "s_sub_u32 %[budget], %[budget], 1",
          "s_cbranch_scc0 1b",
          f"s_andn2_b64 {R}, %[live], %[occ]",
          f"s_bcnt1_i32_b64 {RLO}, {R}",
          f"s_mul_i32 {RLO}, {RLO}, %[thr]",                     # alive * window * share
          f"s_lshl_b32 {RHI}, %[acc], 4",                        # picked up * 16
          "s_mov_b32 %[acc], 0",
          "s_mov_b32 %[budget], %[window]",
          f"s_cmp_lt_u32 {RHI}, {RLO}",
          "s_cbranch_scc0 1b",                                   # coherent enough: go on
          # ---- dissolve: the current node goes back on the stack, every ray continues alone ----------------------------
          "80:",
          "s_mov_b32 m0, %[sp]",
          f"v_writelane_b32 %[vref], {REF}, m0",
          f"v_writelane_b32 %[vlo], {MLO}, m0",
          f"v_writelane_b32 %[vhi], {MHI}, m0",
          "s_add_u32 %[sp], %[sp], 1",
          "s_mov_b32 %[acc], 1",                                 # (the status leaves in acc)
          "s_branch 99f",
          "90:",
          "s_mov_b32 %[acc], 0",
          "s_branch 99f"]
    # ---- out of line: push slot k ----------------------------------------------------------------------------------
    for k in range(4):
        L += [f"4{k}:",
              "s_mov_b32 m0, %[sp]",
              f"v_writelane_b32 %[vref], {N(24 + k)}, m0",
              f"v_writelane_b32 %[vlo], {s(HBASE + 2 * k)}, m0",
              f"v_writelane_b32 %[vhi], {s(HBASE + 1 + 2 * k)}, m0",
              "s_add_u32 %[sp], %[sp], 1"]
        if PREFETCH:
            # the node will be popped later: ask for its two cache lines now (into two registers nobody reads; every
            # later s_waitcnt lgkmcnt(0) covers them), so that the pop finds them in the scalar cache
            L += [f"s_load_dword {s(NSGPR)}, %[wb], {N(24 + k)}",
                  f"s_load_dword {s(NSGPR + 1)}, %[wb], {N(24 + k)} offset:64"]
        L += [f"s_branch 2{k}b"]
    # ---- out of line: leaf slot k ------------------------------------------------------------------------------------
    for k in range(4):
        L += [f"3{k}:",
              # an EMPTY slot (ref END, inverted box) can "hit" in the generic form, which takes max/min of both planes
              # whatever their order; END - 1 must never reach the load below
              f"s_cmp_eq_u32 {N(24 + k)}, -1",
              f"s_cbranch_scc1 2{k}b",
              # the rays that hit this slot's box and are not occluded yet.  (Decided BEFORE the record is requested: scalar
              #  loads return in any order, so a request that nobody waits for could land in T after a later handler's.)
              f"s_andn2_b64 exec, {H(k)}, %[occ]",
              f"s_cbranch_scc0 6{k}f",
              f"s_sub_u32 {REF}, {N(24 + k)}, 1",               # (REF is free here: the node is in registers)
              f"s_load_dwordx8 s[{BASE + 28}:{BASE + 35}], %[tb], {REF}",
              f"s_load_dword {T(8)}, %[tb], {REF} offset:32",
              "s_waitcnt lgkmcnt(0)"]
        L += triangle(R, k)
        L += [f"s_andn2_b64 exec, exec, {R}",                   # lanes whose ray hits the triangle
              f"s_cbranch_scc0 6{k}f"]
        L += exact_box(k, octant, R)                             # ... count iff the ray reaches the leaf: exact test of its parent's box
        L += [f"s_or_b64 %[occ], %[occ], {R}",
              f"6{k}:",
              f"s_branch 2{k}b"]
    L += ["99:",
          "s_waitcnt lgkmcnt(0)",                                # nothing of ours may land in SGPRs after the asm ends
          f"v_readlane_b32 {RLO}, %[vlo], 63",
          f"v_readlane_b32 {RHI}, %[vhi], 63",
          f"s_mov_b64 exec, {R}"]
    return L


def loop_range():
    """The loop of a PIECE of a split tile (rts_kernels.hip, 'SPLIT TILES'): the generic form of loop() with
      * a range filter: the node's dwords 28..31 {own index, first index of slots 1..3} travel in T0..T3 until the filter has
        used them; slot k covers the index range [lo_k, idx_k+1) (lo_0 = own index + 1, a lower bound; an empty slot's index
        is END) and is dropped when it starts at or after %[rb] or ends at or before %[ra];
      * one entry: a POP (the caller pushes the node to start with; the loop pushes back the node it returns on);
      * a return after %[budget] + 1 pops (status 2: the caller looks at what the other pieces found, applies the dissolve
        rule to %[acc] = members picked up, and comes back), on a full stack (1) or when the stack is empty (0).  The status
        leaves in %[budget]."""
    octant = 8
    L = [f"s_mov_b64 {R}, exec",
         f"v_writelane_b32 %[vlo], {RLO}, 63",
         f"v_writelane_b32 %[vhi], {RHI}, 63",
         "s_branch 5f",
         # ---- one node --------------------------------------------------------------------------------------------
         "1:",
         f"s_load_dwordx16 s[{BASE}:{BASE + 15}], %[wb], {REF}",
         f"s_load_dwordx8 s[{BASE + 16}:{BASE + 23}], %[wb], {REF} offset:64",
         f"s_load_dwordx4 s[{BASE + 24}:{BASE + 27}], %[wb], {REF} offset:96",
         f"s_load_dwordx4 s[{BASE + 28}:{BASE + 31}], %[wb], {REF} offset:112",
         "s_cmp_gt_u32 %[sp], 59",
         "s_cbranch_scc1 80f",
         f"s_mov_b32 {NXREF}, -1",
         f"s_mov_b64 exec, {M}",
         "s_waitcnt lgkmcnt(0)"]
    L += cheap_pair(0, 1, octant)
    L += cheap_pair(2, 3, octant)
    L += [f"s_add_u32 {RLO}, {T(0)}, 1"]
    for k in range(4):
        L += [f"s_cmp_ge_u32 {T(k) if k else RLO}, %[rb]",
              f"s_cselect_b64 {H(k)}, 0, {H(k)}"]
        if k < 3:
            L += [f"s_cmp_le_u32 {T(k + 1)}, %[ra]",
                  f"s_cselect_b64 {H(k)}, 0, {H(k)}"]
    for k in range(4):
        L += [f"s_cmp_lg_u64 {H(k)}, 0",
              f"s_cbranch_scc0 2{k}f",
              f"s_bitcmp1_b32 {N(24 + k)}, 0",
              f"s_cbranch_scc1 3{k}f",
              f"s_cmp_eq_u32 {NXREF}, -1",
              f"s_cbranch_scc0 4{k}f",
              f"s_mov_b32 {NXREF}, {N(24 + k)}",
              f"s_mov_b64 {NXM}, {H(k)}",
              f"2{k}:"]
    L += [f"s_cmp_eq_u32 {NXREF}, -1",
          "s_cbranch_scc1 5f",
          f"s_mov_b32 {REF}, {NXREF}",
          f"s_andn2_b64 {M}, {NXM}, %[occ]",
          "s_cbranch_scc1 1b",
          # ---- pop -----------------------------------------------------------------------------------------------
          "5:",
          "s_cmp_eq_u32 %[sp], 0",
          "s_cbranch_scc1 90f",
          "s_sub_u32 %[sp], %[sp], 1",
          "s_mov_b32 m0, %[sp]",
          f"v_readlane_b32 {REF}, %[vref], m0",
          f"v_readlane_b32 {MLO}, %[vlo], m0",
          f"v_readlane_b32 {MHI}, %[vhi], m0",
          f"s_andn2_b64 {M}, {M}, %[occ]",
          "s_cbranch_scc0 5b",
          f"s_bcnt1_i32_b64 {RLO}, {M}",
          f"s_add_u32 %[acc], %[acc], {RLO}",
          "s_sub_u32 %[budget], %[budget], 1",
          "s_cbranch_scc0 1b",
          # ---- the caller's turn: the node just popped goes back on the stack (and is counted when it is popped again) ------
          f"s_sub_u32 %[acc], %[acc], {RLO}",
          "s_mov_b32 m0, %[sp]",
          f"v_writelane_b32 %[vref], {REF}, m0",
          f"v_writelane_b32 %[vlo], {MLO}, m0",
          f"v_writelane_b32 %[vhi], {MHI}, m0",
          "s_add_u32 %[sp], %[sp], 1",
          "s_mov_b32 %[budget], 2",
          "s_branch 99f",
          # ---- stack nearly full: the current node goes back on the stack, every ray continues alone ---------------------
          "80:",
          "s_mov_b32 m0, %[sp]",
          f"v_writelane_b32 %[vref], {REF}, m0",
          f"v_writelane_b32 %[vlo], {MLO}, m0",
          f"v_writelane_b32 %[vhi], {MHI}, m0",
          "s_add_u32 %[sp], %[sp], 1",
          "s_mov_b32 %[budget], 1",
          "s_branch 99f",
          "90:",
          "s_mov_b32 %[budget], 0",
          "s_branch 99f"]
    for k in range(4):
        L += [f"4{k}:",
              "s_mov_b32 m0, %[sp]",
              f"v_writelane_b32 %[vref], {N(24 + k)}, m0",
              f"v_writelane_b32 %[vlo], {s(HBASE + 2 * k)}, m0",
              f"v_writelane_b32 %[vhi], {s(HBASE + 1 + 2 * k)}, m0",
              "s_add_u32 %[sp], %[sp], 1",
              f"s_branch 2{k}b"]
    for k in range(4):
        L += [f"3{k}:",
              f"s_cmp_eq_u32 {N(24 + k)}, -1",
              f"s_cbranch_scc1 2{k}b",
              f"s_andn2_b64 exec, {H(k)}, %[occ]",
              f"s_cbranch_scc0 6{k}f",
              f"s_sub_u32 {REF}, {N(24 + k)}, 1",
              f"s_load_dwordx8 s[{BASE + 28}:{BASE + 35}], %[tb], {REF}",
              f"s_load_dword {T(8)}, %[tb], {REF} offset:32",
              "s_waitcnt lgkmcnt(0)"]
        L += triangle(R, k)
        L += [f"s_andn2_b64 exec, exec, {R}",
              f"s_cbranch_scc0 6{k}f"]
        L += exact_box(k, octant, R)
        L += [f"s_or_b64 %[occ], %[occ], {R}",
              f"6{k}:",
              f"s_branch 2{k}b"]
    L += ["99:",
          "s_waitcnt lgkmcnt(0)",
          f"v_readlane_b32 {RLO}, %[vlo], 63",
          f"v_readlane_b32 {RHI}, %[vhi], 63",
          f"s_mov_b64 exec, {R}"]
    return L


def emit_range(ind):
    lines = loop_range()
    body = "\n".join(f'{ind}    "{l}\\n\\t"' for l in lines)
    outs = ['[sp] "+s"(sp)', '[acc] "+s"(acc)', '[budget] "+s"(budget)', '[occ] "+s"(occ)',
            '[vref] "+v"(stRef)', '[vlo] "+v"(stLo)', '[vhi] "+v"(stHi)']
    outs += [f'[t{i}] "=&v"(t{i})' for i in range(12)]
    ins = ['[wb] "s"(wbase)', '[tb] "s"(tbase)', '[ra] "s"(ra)', '[rb] "s"(rb)']
    ins += [f'[o{a}] "v"(r.o.{a})' for a in AX] + [f'[i{a}] "v"(r.inv.{a})' for a in AX] + [f'[d{a}] "v"(r.d.{a})' for a in AX]
    ins += ['[tm] "s"(tmax)']
    ins += [f'[cu{a}] "v"(w.cU.{a})' for a in AX] + [f'[cd{a}] "v"(w.cD.{a})' for a in AX]
    clob = [f'"s{i}"' for i in range(BASE, BASE + NSGPR)] + ['"vcc"', '"scc"', '"m0"']
    return (f"{ind}asm volatile(\n{body}\n{ind}    : {', '.join(outs)}\n{ind}    : {', '.join(ins)}\n"
            f"{ind}    : {', '.join(clob)});\n")


def emit(octant, ind):
    lines = loop(octant)
    body = "\n".join(f'{ind}    "{l}\\n\\t"' for l in lines)
    outs = ['[sp] "+s"(sp)', '[acc] "+s"(acc)', '[budget] "+s"(budget)', '[occ] "+s"(occ)',
            '[vref] "+v"(stRef)', '[vlo] "+v"(stLo)', '[vhi] "+v"(stHi)']
    outs += [f'[t{i}] "=&v"(t{i})' for i in range(12)]
    ins = ['[wb] "s"(wbase)', '[tb] "s"(tbase)', '[live] "s"(live)', '[thr] "s"(thr)', '[window] "s"(window)']
    ins += [f'[o{a}] "v"(r.o.{a})' for a in AX] + [f'[i{a}] "v"(r.inv.{a})' for a in AX] + [f'[d{a}] "v"(r.d.{a})' for a in AX]
    ins += ['[tm] "s"(tmax)']
    ins += [f'[cu{a}] "v"(w.cU.{a})' for a in AX] + [f'[cd{a}] "v"(w.cD.{a})' for a in AX]
    clob = [f'"s{i}"' for i in range(BASE, BASE + NSGPR + (2 if PREFETCH else 0))] + ['"vcc"', '"scc"', '"m0"']
    return (f"{ind}asm volatile(\n{body}\n{ind}    : {', '.join(outs)}\n{ind}    : {', '.join(ins)}\n"
            f"{ind}    : {', '.join(clob)});\n")


def main():
    o = ["// GENERATED by tools/gen_wide_asm.py -- do not edit by hand.",
         "// The loop of the wide packet kernel (see rts_kernels.hip, 'V_WIDE').  Walks the private wide nodes from the root with",
         "// the stack in stRef/stLo/stHi (entry i in lane i).  Returns 0 when the packet is finished and 1 when it dissolves",
         "// (too few members per popped node, or the stack nearly full): every pending (node, members) is on the stack then.",
         "// form 0..7: ordered slab tests for the sign octant (bit a set <=> 1/d component a negative in every lane); 8: generic.",
         "",
         "__device__ __forceinline__ uint32_t wideDescend(uint32_t form, const void* wbase, const void* tbase, const Ray& r,",
         "                                              const WideRay& w, float tmax, uint64_t live, uint64_t& occ, uint32_t& sp,",
         "                                              uint32_t& stRef, uint32_t& stLo, uint32_t& stHi, uint32_t window, uint32_t thr) {",
         "    uint32_t acc = 0;",
         "    int32_t budget = (int32_t)window - 1;",
         "    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;",
         "    switch (form) {"]
    for octant in range(9):
        o.append(f"    case {octant}:" if octant < 8 else "    default:")
        o.append(emit(octant, "        ").rstrip("\n"))
        o.append("        break;")
    o += ["    }", "    return acc;", "}", "",
          "// The same walk for a PIECE of a split tile (generic form): pops from the stack the caller prepared, drops the slots whose",
          "// index range lies outside [ra, rb), and returns after `pops` pops (2), on a full stack (1) or when the stack is empty (0);",
          "// `acc` keeps adding up the members picked up per pop.",
          "__device__ __forceinline__ uint32_t wideDescendRange(const void* wbase, const void* tbase, const Ray& r, const WideRay& w, float tmax,",
          "                                                   uint32_t ra, uint32_t rb, uint64_t& occ, uint32_t& sp, uint32_t& stRef, uint32_t& stLo,",
          "                                                   uint32_t& stHi, uint32_t& acc, uint32_t pops) {",
          "    int32_t budget = (int32_t)pops - 1;",
          "    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;",
          emit_range("    ").rstrip("\n"),
          "    return (uint32_t)budget;", "}", ""]
    open(OUT, "w").write("\n".join(o))
    print("wrote", OUT, sum(1 for _ in open(OUT)), "lines")


if __name__ == "__main__":
    main()
