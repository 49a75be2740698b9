#!/usr/bin/env python3
"""Generates raytracedshadows_amd/csrc/rts_packet_asm.inc: the hand-written gfx950 descent loop of the
packet kernels, instantiated for K = 1, 2, 4 ray sets per lane and 9 slab-test forms (8 sign octants
of the "ordered" test + the generic min/max test).  The .inc file is committed; re-run this script
after editing the loop:

    python tools/gen_packet_asm.py

Register plan inside the asm (fixed scratch SGPRs, declared as clobbers):
    s[40:47]  node: s40-42 bboxMin, s43 leaf tag, s44-46 bboxMax, s47 miss link
    s[48:49]  saved EXEC          s[50:51]  leavers of the set being processed
    s52       byte offset of the node (cur * 32)
    s[54:61]  slab-test results of ray sets 0..3 (VALU-written, SALU-read: interlocked by hardware)
"""
import os
import re
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "raytracedshadows_amd", "csrc", "rts_packet_asm.inc")

SGPR_BASE = 40   # first of the 24 fixed scratch SGPRs (the text below is written for 40 and relocated)

LO = ["s40", "s41", "s42"]
HI = ["s44", "s45", "s46"]
AX = "xyz"


def test_ordered(k, octant):
    """16 VALU: far/near planes picked by the (wave-uniform) sign of 1/d per axis."""
    far = [LO[a] if (octant >> a) & 1 else HI[a] for a in range(3)]
    near = [HI[a] if (octant >> a) & 1 else LO[a] for a in range(3)]
    L = []
    for i, a in enumerate(AX):
        L.append(f"v_sub_f32 %[t{i}], {far[i]}, %[o{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_sub_f32 %[t{3 + i}], {near[i]}, %[o{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_mul_f32 %[t{i}], %[t{i}], %[i{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_mul_f32 %[t{3 + i}], %[t{3 + i}], %[i{a}{k}]")
    L += ["v_min3_f32 %[t0], %[t0], %[t1], %[t2]",
          "v_max_f32 %[t3], %[t3], %[t4]",
          "v_max3_f32 %[t3], %[t3], %[t5], 0",
          f"v_cmp_ge_f32 s[{54 + 2 * k}:{55 + 2 * k}], %[t0], %[t3]"]
    return L


def test_generic(k):
    """22 VALU: the FAST form of boxHit (v_min/v_max per axis), no assumption on signs or box order."""
    L = []
    for i, a in enumerate(AX):
        L.append(f"v_sub_f32 %[t{i}], {HI[i]}, %[o{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_sub_f32 %[t{3 + i}], {LO[i]}, %[o{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_mul_f32 %[t{i}], %[t{i}], %[i{a}{k}]")
    for i, a in enumerate(AX):
        L.append(f"v_mul_f32 %[t{3 + i}], %[t{3 + i}], %[i{a}{k}]")
    L += ["v_max_f32 %[t6], %[t0], %[t3]", "v_min_f32 %[t0], %[t0], %[t3]",
          "v_max_f32 %[t3], %[t1], %[t4]", "v_min_f32 %[t1], %[t1], %[t4]",
          "v_max_f32 %[t4], %[t2], %[t5]", "v_min_f32 %[t2], %[t2], %[t5]",
          "v_min3_f32 %[t6], %[t6], %[t3], %[t4]",
          "v_max_f32 %[t0], %[t0], %[t1]",
          "v_max3_f32 %[t0], %[t0], %[t2], 0",
          f"v_cmp_ge_f32 s[{54 + 2 * k}:{55 + 2 * k}], %[t6], %[t0]"]
    return L


# Every loop starts with the same two instructions: a packet that has no node left (cur == END) must never reach the
# node load -- END * 32 wraps to byte offset 0xFFFFFFE0, 4 GiB - 32 B past the stream (the one scalar-load fault on
# record).  All callers already guarantee cur != END; the guard makes it a property of the loop itself.
ENTRY_GUARD = ["s_cmp_eq_u32 %[cur], -1",
               "s_cbranch_scc1 6f"]


def loop(K, form):
    L = ENTRY_GUARD + [
         "s_lshl_b32 s52, %[cur], 5",
         "s_branch 2f",
         "5:",
         "s_add_u32 s52, s52, 32",                      # down-step: the left child is the next node in memory
         "2:",
         "s_load_dwordx8 s[40:47], %[base], s52",
         "s_waitcnt lgkmcnt(0)",
         "s_cmp_lg_u32 s43, -1",
         "s_cbranch_scc1 9f"]                            # leaf: leave with cur = this node
    for k in range(K):
        L += test_generic(k) if form == 8 else test_ordered(k, form)
    for k in range(K):
        r = f"s[{54 + 2 * k}:{55 + 2 * k}]"
        L += [f"s_andn2_b64 s[50:51], %[m{k}], {r}",   # members whose own test failed: they leave the packet ...
              # (K == 1: nobody leaves => the membership is unchanged and non-empty: straight to the next node)
              "s_cbranch_scc0 5b" if K == 1 else f"s_cbranch_scc0 1{k}f",
              "s_mov_b64 s[48:49], exec",
              "s_mov_b64 exec, s[50:51]",
              f"v_mov_b32 %[w{k}], s47",                # ... and wait on the miss link
              "s_mov_b64 exec, s[48:49]",
              f"1{k}:",
              f"s_and_b64 %[m{k}], %[m{k}], {r}"]      # the rest goes down
    if K == 2:
        L.append("s_or_b64 s[50:51], %[m0], %[m1]")
    elif K == 4:
        L += ["s_or_b64 s[50:51], %[m0], %[m1]", "s_or_b64 s[48:49], %[m2], %[m3]",
              "s_or_b64 s[50:51], s[50:51], s[48:49]"]
    L += ["s_cbranch_scc1 5b",                          # (K == 1: SCC still comes from the s_and above)
          "s_lshl_b32 s52, s47, 5",                     # side-step: nobody entered the subtree
          "s_cmp_eq_u32 s47, -1",
          "s_cbranch_scc1 6f"]
    for k in range(K):
        L.append(f"v_cmp_eq_u32 %[m{k}], s47, %[w{k}]")  # whoever waits on that node is the packet now
    for k in range(K):                                  # coherence statistics: rays picked up by this side-step
        L += [f"s_bcnt1_i32_b64 s50, %[m{k}]", "s_add_u32 %[acc], %[acc], s50"]
    L += ["s_sub_u32 %[budget], %[budget], 1",
          "s_cbranch_scc0 2b"]
    # a window of side-steps is over: rays picked up per side-step against rays alive (dissolve rule)
    for k in range(K):
        L += [f"v_cmp_ne_u32 s[48:49], -1, %[w{k}]",     # rays of set k that wait somewhere ...
              f"s_or_b64 s[48:49], s[48:49], %[m{k}]",   # ... or walk with the packet
              "s_bcnt1_i32_b64 s50, s[48:49]",
              "s_add_u32 s51, s51, s50" if k else "s_mov_b32 s51, s50"]
    L += ["s_mul_i32 s51, s51, %[thr]",                  # alive * window * share
          "s_lshl_b32 s50, %[acc], 4",                    # picked up * 16
          "s_mov_b32 %[acc], 0",
          "s_mov_b32 %[budget], %[window]",
          "s_cmp_lt_u32 s50, s51",
          "s_cbranch_scc0 2b",                            # coherent enough: go on
          "8:",                                           # dissolve: continue lane-per-ray at s52
          "s_mov_b32 %[leaf], 0",
          "s_lshr_b32 %[cur], s52, 5",
          "s_branch 7f",
          "6:",                                           # miss link END: finished
          "s_mov_b32 %[leaf], 0",
          "s_mov_b32 %[cur], -1",
          "s_branch 7f",
          "9:",
          "s_mov_b32 %[leaf], 1",
          "s_lshr_b32 %[cur], s52, 5",
          "7:"]
    return L


def loop_prefetch(form):
    """K = 1 with the sequential successor prefetched: s[40:47] = node `cur`, s[56:63] = node cur+1 (the down-step
    target), loaded while `cur` is being tested.  Scalar loads may return out of order, so every wait is
    lgkmcnt(0), and the prefetch is drained before its registers are re-targeted or the asm is left."""
    k = 0
    test = test_generic(k) if form == 8 else test_ordered(k, form)
    r = "s[54:55]"
    L = ENTRY_GUARD + [
         "s_lshl_b32 s52, %[cur], 5",
         "2:",                                           # (re)load: current node + its sequential successor
         "s_load_dwordx8 s[40:47], %[base], s52",
         "s_add_u32 s53, s52, 32",
         "s_load_dwordx8 s[56:63], %[base], s53",
         "s_waitcnt lgkmcnt(0)",
         "3:",                                           # s[40:47] valid; a prefetch into s[56:63] may be in flight
         "s_cmp_lg_u32 s43, -1",
         "s_cbranch_scc1 9f"]
    L += test
    L += [f"s_andn2_b64 s[50:51], %[m0], {r}",
          "s_cbranch_scc0 10f",
          "s_mov_b64 s[48:49], exec",
          "s_mov_b64 exec, s[50:51]",
          "v_mov_b32 %[w0], s47",
          "s_mov_b64 exec, s[48:49]",
          "10:",
          f"s_and_b64 %[m0], %[m0], {r}",
          "s_cbranch_scc0 4f",
          # down-step: the successor is (being) loaded already
          "s_add_u32 s52, s52, 32",
          "s_waitcnt lgkmcnt(0)",
          "s_mov_b64 s[40:41], s[56:57]",
          "s_mov_b64 s[42:43], s[58:59]",
          "s_mov_b64 s[44:45], s[60:61]",
          "s_mov_b64 s[46:47], s[62:63]",
          "s_add_u32 s53, s52, 32",
          "s_load_dwordx8 s[56:63], %[base], s53",      # prefetch the next one; overlaps the test of this node
          "s_branch 3b",
          "4:",                                          # side-step
          "s_waitcnt lgkmcnt(0)",                        # drain the stale prefetch before re-targeting s[56:63]
          "s_lshl_b32 s52, s47, 5",
          "s_cmp_eq_u32 s47, -1",
          "s_cbranch_scc1 6f",
          "v_cmp_eq_u32 %[m0], s47, %[w0]",
          "s_bcnt1_i32_b64 s50, %[m0]",
          "s_add_u32 %[acc], %[acc], s50",
          "s_sub_u32 %[budget], %[budget], 1",
          "s_cbranch_scc0 2b",
          "v_cmp_ne_u32 s[48:49], -1, %[w0]",
          "s_or_b64 s[48:49], s[48:49], %[m0]",
          "s_bcnt1_i32_b64 s51, s[48:49]",
          "s_mul_i32 s51, s51, %[thr]",
          "s_lshl_b32 s50, %[acc], 4",
          "s_mov_b32 %[acc], 0",
          "s_mov_b32 %[budget], %[window]",
          "s_cmp_lt_u32 s50, s51",
          "s_cbranch_scc0 2b",
          "8:",
          "s_mov_b32 %[leaf], 0",
          "s_lshr_b32 %[cur], s52, 5",
          "s_branch 7f",
          "6:",
          "s_mov_b32 %[leaf], 0",
          "s_mov_b32 %[cur], -1",
          "s_branch 7f",
          "9:",
          "s_mov_b32 %[leaf], 1",
          "s_lshr_b32 %[cur], s52, 5",
          "7:",
          "s_waitcnt lgkmcnt(0)"]                        # nothing of ours may land in SGPRs after the asm ends
    return L



def loop_leaf(form):
    """K = 1, leaves handled inside the loop (no exit to compiled code per leaf visit).

    Triangle test = RayTracedShadows.comp:41-59 operation for operation (separate mul/sub/add, no contraction;
    dot products left to right; 1/det with the correctly rounded divide sequence hipcc emits for `1.0f / x`).
    Extra fixed SGPRs: s[56:59] = tail vec4 (v0), s[60:63] = reject masks.  Returns 0 (finished: cur = END, or
    dissolve: cur = node to continue at) or 2 (every ray on the leaf got occluded and nobody waits on its miss
    link: the caller finds the lowest waiting node)."""
    test = test_generic(0) if form == 8 else test_ordered(0, form)
    r = "s[54:55]"
    L = ENTRY_GUARD + [
         "s_lshl_b32 s52, %[cur], 5",
         "s_branch 2f",
         "5:",
         "s_add_u32 s52, s52, 32",
         "2:",
         "s_load_dwordx8 s[40:47], %[base], s52",
         "s_waitcnt lgkmcnt(0)",
         "s_cmp_lg_u32 s43, -1",
         "s_cbranch_scc1 9f"]
    L += test
    L += [f"s_andn2_b64 s[50:51], %[m0], {r}",            # members whose own test failed: they leave the packet ...
          "s_cbranch_scc0 5b",
          "v_mov_b32 %[t0], s47",
          "v_cndmask_b32_e64 %[w0], %[w0], %[t0], s[50:51]",   # ... and wait on the miss link
          f"s_and_b64 %[m0], %[m0], {r}",
          "s_cbranch_scc1 5b",
          "4:",                                           # side-step to s47
          "s_lshl_b32 s52, s47, 5",
          "s_cmp_eq_u32 s47, -1",
          "s_cbranch_scc1 6f",
          "v_cmp_eq_u32 %[m0], s47, %[w0]",
          "s_bcnt1_i32_b64 s50, %[m0]",
          "s_add_u32 %[acc], %[acc], s50",
          "s_sub_u32 %[budget], %[budget], 1",
          "s_cbranch_scc0 2b",
          "v_cmp_ne_u32 s[48:49], -1, %[w0]",
          "s_or_b64 s[48:49], s[48:49], %[m0]",
          "s_bcnt1_i32_b64 s51, s[48:49]",
          "s_mul_i32 s51, s51, %[thr]",
          "s_lshl_b32 s50, %[acc], 4",
          "s_mov_b32 %[acc], 0",
          "s_mov_b32 %[budget], %[window]",
          "s_cmp_lt_u32 s50, s51",
          "s_cbranch_scc0 2b",
          "8:",                                           # dissolve
          "s_mov_b32 %[leaf], 0",
          "s_lshr_b32 %[cur], s52, 5",
          "s_branch 7f",
          "6:",                                           # finished
          "s_mov_b32 %[leaf], 0",
          "s_mov_b32 %[cur], -1",
          "s_branch 7f",
          # ---- leaf: e0 = s40-42, tail index s43, e1 = s44-46, next = s47 -------------------------------------
          "9:",
          "s_lshl_b32 s53, s43, 4",
          "s_load_dwordx4 s[56:59], %[base], s53",        # v0 (arrives while s1, det and 1/det are computed)
          "v_mul_f32 %[t0], s46, %[dy0]", "v_mul_f32 %[t1], s45, %[dz0]", "v_sub_f32 %[t0], %[t0], %[t1]",   # s1.x = d.y*e1.z - e1.y*d.z
          "v_mul_f32 %[t1], s44, %[dz0]", "v_mul_f32 %[t2], s46, %[dx0]", "v_sub_f32 %[t1], %[t1], %[t2]",   # s1.y = d.z*e1.x - e1.z*d.x
          "v_mul_f32 %[t2], s45, %[dx0]", "v_mul_f32 %[t3], s44, %[dy0]", "v_sub_f32 %[t2], %[t2], %[t3]",   # s1.z = d.x*e1.y - e1.x*d.y
          "v_mul_f32 %[t3], s40, %[t0]", "v_mul_f32 %[t4], s41, %[t1]", "v_add_f32 %[t3], %[t3], %[t4]",
          "v_mul_f32 %[t4], s42, %[t2]", "v_add_f32 %[t3], %[t3], %[t4]",                                     # det = dot(s1, e0)
          # invd = 1.0f / det, correctly rounded.  Where every lane's det has a biased exponent in 27..226 (2^-100 <= |det| <
          # 2^100): v_rcp_f32 + one Newton step, which IS the IEEE quotient there (every bit pattern checked on the device:
          # rts_selftest_reciprocal, rts_kernels.hip: rcpFast); otherwise the general division (the sequence hipcc emits).
          "v_bfe_u32 %[t5], %[t3], 23, 8",
          "v_subrev_u32 %[t5], 27, %[t5]",
          "v_cmp_lt_u32 vcc, 0xc7, %[t5]",                # lanes out of the range (unsigned: also exponents below 27)
          "s_cbranch_vccnz 11f",
          "v_rcp_f32 %[t4], %[t3]",
          "s_nop 0",                                     # (gfx940+: a VALU op that reads a transcendental's result needs one wait state; the assembler adds none)
          "v_fma_f32 %[t5], -%[t3], %[t4], 1.0",
          "v_fma_f32 %[t4], %[t5], %[t4], %[t4]",
          "s_branch 12f",
          "11:",
          "v_div_scale_f32 %[t5], s[60:61], %[t3], %[t3], 1.0",
          "v_rcp_f32 %[t7], %[t5]",
          "v_div_scale_f32 %[t6], vcc, 1.0, %[t3], 1.0",
          "v_fma_f32 %[t8], -%[t5], %[t7], 1.0",
          "v_fmac_f32 %[t7], %[t8], %[t7]",
          "v_mul_f32 %[t8], %[t6], %[t7]",
          "v_fma_f32 %[t9], -%[t5], %[t8], %[t6]",
          "v_fmac_f32 %[t8], %[t9], %[t7]",
          "v_fma_f32 %[t5], -%[t5], %[t8], %[t6]",
          "v_div_fmas_f32 %[t5], %[t5], %[t7], %[t8]",
          "v_div_fixup_f32 %[t4], %[t5], %[t3], 1.0",                                                          # invd
          "12:",
          "s_waitcnt lgkmcnt(0)",
          "v_subrev_f32 %[t5], s56, %[ox0]", "v_subrev_f32 %[t6], s57, %[oy0]", "v_subrev_f32 %[t7], s58, %[oz0]",   # dd = o - v0
          "v_mul_f32 %[t8], %[t5], %[t0]", "v_mul_f32 %[t9], %[t6], %[t1]", "v_add_f32 %[t8], %[t8], %[t9]",
          "v_mul_f32 %[t9], %[t7], %[t2]", "v_add_f32 %[t8], %[t8], %[t9]", "v_mul_f32 %[t8], %[t8], %[t4]", # b1 = dot(dd, s1) * invd
          # reject = b1<0 || b1>1 || b2<0 || b1+b2>1 || t<0 || t>tmax   (ordered compares: false on NaN, comp:51).  A disjunction:
          # once EVERY member is rejected by the conditions evaluated so far the rest of the test cannot change anything and
          # is skipped (courtyard: every ray fails on b1 alone in 41 % of the wave-wide tests, before t in 69 %: tools/wide_sim.py)
          "v_cmp_gt_f32 s[60:61], 0, %[t8]",
          "v_cmp_lt_f32 s[62:63], 1.0, %[t8]", "s_or_b64 s[60:61], s[60:61], s[62:63]",
          "s_andn2_b64 s[62:63], %[m0], s[60:61]",
          "s_cbranch_scc0 13f",
          "v_mul_f32 %[t9], s42, %[t6]", "v_mul_f32 %[t10], s41, %[t7]", "v_sub_f32 %[t9], %[t9], %[t10]",     # s2.x = dd.y*e0.z - e0.y*dd.z
          "v_mul_f32 %[t10], s40, %[t7]", "v_mul_f32 %[t11], s42, %[t5]", "v_sub_f32 %[t10], %[t10], %[t11]",  # s2.y = dd.z*e0.x - e0.z*dd.x
          "v_mul_f32 %[t11], s41, %[t5]", "v_mul_f32 %[t12], s40, %[t6]", "v_sub_f32 %[t11], %[t11], %[t12]",  # s2.z = dd.x*e0.y - e0.x*dd.y
          "v_mul_f32 %[t12], %[dx0], %[t9]", "v_mul_f32 %[t13], %[dy0], %[t10]", "v_add_f32 %[t12], %[t12], %[t13]",
          "v_mul_f32 %[t13], %[dz0], %[t11]", "v_add_f32 %[t12], %[t12], %[t13]", "v_mul_f32 %[t12], %[t12], %[t4]",   # b2 = dot(d, s2) * invd
          "v_add_f32 %[t14], %[t8], %[t12]",                                                                            # b1 + b2
          "v_cmp_gt_f32 s[62:63], 0, %[t12]", "s_or_b64 s[60:61], s[60:61], s[62:63]",
          "v_cmp_lt_f32 s[62:63], 1.0, %[t14]", "s_or_b64 s[60:61], s[60:61], s[62:63]",
          "s_andn2_b64 s[62:63], %[m0], s[60:61]",
          "s_cbranch_scc0 13f",
          "v_mul_f32 %[t13], s44, %[t9]", "v_mul_f32 %[t14], s45, %[t10]", "v_add_f32 %[t13], %[t13], %[t14]",
          "v_mul_f32 %[t14], s46, %[t11]", "v_add_f32 %[t13], %[t13], %[t14]", "v_mul_f32 %[t13], %[t13], %[t4]",     # t = dot(e1, s2) * invd
          "v_cmp_gt_f32 s[62:63], 0, %[t13]", "s_or_b64 s[60:61], s[60:61], s[62:63]",
          "v_cmp_lt_f32 s[62:63], %[tm0], %[t13]", "s_or_b64 s[60:61], s[60:61], s[62:63]",
          "13:",
          "s_andn2_b64 s[62:63], %[m0], s[60:61]",         # members that hit the triangle: occluded, finished
          "s_or_b64 %[oc0], %[oc0], s[62:63]",
          "s_and_b64 s[60:61], %[m0], s[60:61]",           # members that missed: wait on the miss link
          "s_mov_b64 s[48:49], exec",
          "s_mov_b64 exec, s[60:61]",
          "v_mov_b32 %[w0], s47",
          "s_mov_b64 exec, s[62:63]",
          "v_mov_b32 %[w0], -1",
          "s_mov_b64 exec, s[48:49]",
          "s_lshl_b32 s52, s47, 5",
          "s_cmp_eq_u32 s47, -1",
          "s_cbranch_scc1 6b",
          "v_cmp_eq_u32 %[m0], s47, %[w0]",
          "s_cmp_eq_u64 %[m0], 0",
          "s_cbranch_scc0 2b",                             # somebody stands on the miss link: go on
          "s_mov_b32 %[leaf], 2",                          # nobody: the caller looks for the lowest waiting node
          "s_lshr_b32 %[cur], s52, 5",
          "7:"]
    return L


def relocate(line):
    """Moves every fixed scratch register s40..s63 (single or range) to SGPR_BASE.."""
    d = SGPR_BASE - 40
    def one(m):
        return f"s{int(m.group(1)) + d}"
    def rng(m):
        return f"s[{int(m.group(1)) + d}:{int(m.group(2)) + d}]"
    line = re.sub(r"\bs\[(4\d|5\d|6[0-3]):(4\d|5\d|6[0-3])\]", rng, line)
    return re.sub(r"\bs(4\d|5\d|6[0-3])\b", one, line)


def emit_asm(K, form, ind, prefetch=False, leaf=False):
    lines = loop_leaf(form) if leaf else (loop_prefetch(form) if prefetch else loop(K, form))
    lines = [relocate(l) for l in lines]
    body = "\n".join(f'{ind}    "{l}\\n\\t"' for l in lines)
    outs = ['[cur] "+s"(cur)', '[budget] "+s"(budget)', '[acc] "+s"(acc)', '[leaf] "=&s"(leaf)']
    outs += [f'[m{k}] "+s"(members[{k}])' for k in range(K)]
    outs += [f'[w{k}] "+v"(wait[{k}])' for k in range(K)]
    outs += [f'[t{i}] "=&v"(t{i})' for i in range(15 if leaf else 7)]
    if leaf:
        outs += ['[oc0] "+s"(occluded[0])']
    ins = ['[base] "s"(base)', '[thr] "s"(thr)', '[window] "s"(window)']
    for k in range(K):
        ins += [f'[o{a}{k}] "v"(r[{k}].o.{a})' for a in AX] + [f'[i{a}{k}] "v"(r[{k}].inv.{a})' for a in AX]
        if leaf:
            ins += [f'[d{a}{k}] "v"(r[{k}].d.{a})' for a in AX] + [f'[tm{k}] "v"(r[{k}].tmax)']
    regs = list(range(40, 64)) if (prefetch or leaf) else list(range(40, 53)) + list(range(54, 54 + 2 * K))
    clob = [f'"s{i + SGPR_BASE - 40}"' for i in regs] + ['"vcc"', '"scc"']
    return (f"{ind}asm volatile(\n{body}\n{ind}    : {', '.join(outs)}\n{ind}    : {', '.join(ins)}\n"
            f"{ind}    : {', '.join(clob)});\n")


def main():
    global SGPR_BASE
    if len(sys.argv) > 1:
        SGPR_BASE = int(sys.argv[1])
    o = ["// GENERATED by tools/gen_packet_asm.py -- do not edit by hand.",
         "// Descent loop of the packet kernels (see rts_kernels.hip, 'V_PACKET').  Walks inner nodes from `cur`",
         "// until the packet stands on a leaf (returns 1, cur = that leaf), runs out of nodes (returns 0, cur = END)",
         "// or decides to dissolve (returns 0, cur = node to continue at).  Dissolve rule: every `window`+1 side-steps",
         "// the rays picked up at those side-steps (`acc`) are compared with the rays alive: acc*16 < alive*thr with",
         "// thr = (window+1)*share dissolves the packet.",
         "// form 0..7: ordered slab test for the sign octant (bit a set <=> 1/d component a negative in every lane);",
         "// form 8: generic FAST slab test.", ""]
    for K in (1, 2, 4):
        o.append(f"__device__ __forceinline__ uint32_t packetDescend(uint32_t form, const void* base, const Ray (&r)[{K}],")
        o.append(f"                                                uint32_t& cur, uint64_t (&members)[{K}], uint32_t (&wait)[{K}],")
        o.append("                                                int32_t& budget, uint32_t& acc, uint32_t thr, uint32_t window) {")
        o.append("    uint32_t leaf;")
        o.append("    float t0, t1, t2, t3, t4, t5, t6;")
        o.append("    switch (form) {")
        for form in range(9):
            o.append(f"    case {form}:" if form < 8 else "    default:")
            o.append(emit_asm(K, form, "        ").rstrip("\n"))
            o.append("        break;")
        o.append("    }")
        o.append("    return leaf;")
        o.append("}")
        o.append("")
    o.append("// K = 1 with the leaves handled inside the loop (see loop_leaf in the generator).")
    o.append("__device__ __forceinline__ uint32_t packetDescendLeaf(uint32_t form, const void* base, const Ray (&r)[1],")
    o.append("                                                    uint32_t& cur, uint64_t (&members)[1], uint32_t (&wait)[1],")
    o.append("                                                    uint64_t (&occluded)[1], int32_t& budget, uint32_t& acc,")
    o.append("                                                    uint32_t thr, uint32_t window) {")
    o.append("    uint32_t leaf;")
    o.append("    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14;")
    o.append("    switch (form) {")
    for form in range(9):
        o.append(f"    case {form}:" if form < 8 else "    default:")
        o.append(emit_asm(1, form, "        ", leaf=True).rstrip("\n"))
        o.append("        break;")
    o.append("    }")
    o.append("    return leaf;")
    o.append("}")
    o.append("")
    o.append("// K = 1 with the sequential successor node prefetched into a second SGPR set (see loop_prefetch).")
    o.append("__device__ __forceinline__ uint32_t packetDescendPrefetch(uint32_t form, const void* base, const Ray (&r)[1],")
    o.append("                                                        uint32_t& cur, uint64_t (&members)[1], uint32_t (&wait)[1],")
    o.append("                                                        int32_t& budget, uint32_t& acc, uint32_t thr, uint32_t window) {")
    o.append("    uint32_t leaf;")
    o.append("    float t0, t1, t2, t3, t4, t5, t6;")
    o.append("    switch (form) {")
    for form in range(9):
        o.append(f"    case {form}:" if form < 8 else "    default:")
        o.append(emit_asm(1, form, "        ", prefetch=True).rstrip("\n"))
        o.append("        break;")
    o.append("    }")
    o.append("    return leaf;")
    o.append("}")
    o.append("")
    open(OUT, "w").write("\n".join(o))
    print("wrote", OUT, sum(1 for _ in open(OUT)), "lines")


if __name__ == "__main__":
    main()
