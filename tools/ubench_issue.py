#!/usr/bin/env python3
"""Instruction-issue micro-benchmark for gfx950 (development aid, not product): what does ONE vector instruction of each kind
cost a SIMD, alone and beside v_mfma_f32_32x32x16_bf16, at 1..8 waves per SIMD?  Round 4: the exact scoring epilogue is 40
vector operations per 32 x 32 step beside two MFMAs and the kernel's time follows the operation count, so the table this
prints is what an epilogue has to be designed against.

    python tools/ubench_issue.py            # writes tools/gen/ubench_issue.hip and builds tools/ubench_issue.bin (hipcc, no GPU needed)
    tools/ubench_issue.bin [filter]         # on the MI355X

Every kernel is one inline-asm loop over hard-coded registers (the compiler schedules nothing); a workgroup is 4 waves = one
wave per SIMD, W workgroups per CU are forced by the dynamic-LDS size, the grid fills every CU.  A wave reads s_memtime
(shader clock) and s_memrealtime (100 MHz) around its loop; the SIMD has completed W * iters iterations in that time:
cycles per iteration per SIMD = cycles / (W * iters).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "gen")

# (name, asm template): {d} destination, {a} {b} {c} sources (32-bit VGPRs); {d2} {a2} {b2} 64-bit aligned pairs
OPS = [
    ("v_fma_f32", "v_fma_f32 {d}, {a}, {b}, {c}"),
    ("v_add_f32_e32", "v_add_f32_e32 {d}, {a}, {b}"),
    ("v_mul_f32_e32", "v_mul_f32_e32 {d}, {a}, {b}"),
    ("v_min_f32_e32", "v_min_f32_e32 {d}, {a}, {b}"),
    ("v_min_f32_e64", "v_min_f32_e64 {d}, {a}, {b}"),
    ("v_min_f32_e64_abs", "v_min_f32_e64 {d}, |{a}|, |{b}|"),
    ("v_min_f32_e64_clamp", "v_min_f32_e64 {d}, {a}, {b} clamp"),
    ("v_sub_f32_abs_clamp", "v_sub_f32_e64 {d}, {a}, |{b}| clamp"),
    ("v_min3_f32_const1", "v_min3_f32 {d}, {a}, {b}, 1.0"),
    ("v_min3_f32", "v_min3_f32 {d}, {a}, {b}, {c}"),
    ("v_min3_f32_abs", "v_min3_f32 {d}, {c}, |{a}|, |{b}|"),
    ("v_med3_f32", "v_med3_f32 {d}, {a}, {b}, {c}"),
    ("v_max3_f32", "v_max3_f32 {d}, {a}, {b}, {c}"),
    ("v_cvt_pknorm_u16_f32", "v_cvt_pknorm_u16_f32 {d}, {a}, {b}"),
    ("v_cvt_pknorm_i16_f32", "v_cvt_pknorm_i16_f32 {d}, {a}, {b}"),
    ("v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 {d}, {a}, {b}"),
    ("v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 {d}, {a}, {b}"),
    ("v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 {d}, {a}, 1, {c}"),
    ("v_cvt_pk_u16_u32", "v_cvt_pk_u16_u32 {d}, {a}, {b}"),
    ("v_cvt_pk_i16_i32", "v_cvt_pk_i16_i32 {d}, {a}, {b}"),
    ("v_cvt_i32_f32", "v_cvt_i32_f32_e32 {d}, {a}"),
    ("v_add3_u32", "v_add3_u32 {d}, {a}, {b}, {c}"),
    ("v_add_u32_e32", "v_add_u32_e32 {d}, {a}, {b}"),
    ("v_sad_u16", "v_sad_u16 {d}, {a}, {b}, {c}"),
    ("v_sad_u8", "v_sad_u8 {d}, {a}, {b}, {c}"),
    ("v_sad_u32", "v_sad_u32 {d}, {a}, {b}, {c}"),
    ("v_msad_u8", "v_msad_u8 {d}, {a}, {b}, {c}"),
    ("v_pk_min_i16", "v_pk_min_i16 {d}, {a}, {b}"),
    ("v_pk_min_u16", "v_pk_min_u16 {d}, {a}, {b}"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {a}, {b}"),
    ("v_pk_add_i16_clamp", "v_pk_add_i16 {d}, {a}, {b} clamp"),
    ("v_pk_min_f16", "v_pk_min_f16 {d}, {a}, {b}"),
    ("v_pk_add_f16", "v_pk_add_f16 {d}, {a}, {b}"),
    ("v_pk_fma_f16", "v_pk_fma_f16 {d}, {a}, {b}, {c}"),
    ("v_pk_add_f32", "v_pk_add_f32 {d2}, {a2}, {b2}"),
    ("v_pk_mul_f32", "v_pk_mul_f32 {d2}, {a2}, {b2}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {d2}, {a2}, {b2}, {a2}"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {d}, {a}, {b}, {c}"),
    ("v_dot2_i32_i16", "v_dot2_i32_i16 {d}, {a}, {b}, {c}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {d}, {a}, {b}, {c}"),
    ("v_dot2_f32_f16", "v_dot2_f32_f16 {d}, {a}, {b}, {c}"),
    ("v_dot2c_f32_bf16", "v_dot2c_f32_bf16_e32 {d}, {a}, {b}"),
    ("v_fma_mixlo_f16", "v_fma_mixlo_f16 {d}, {a}, 1.0, {b}"),
    ("v_perm_b32", "v_perm_b32 {d}, {a}, {b}, {c}"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {a}, {b}, {c}"),
    ("v_or3_b32", "v_or3_b32 {d}, {a}, {b}, {c}"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {a}, {b}, {c}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {a}, 1, {c}"),
    ("v_xad_u32", "v_xad_u32 {d}, {a}, {b}, {c}"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {a}, {b}, 31"),
    ("v_and_b32_e32", "v_and_b32_e32 {d}, {a}, {b}"),
    ("v_cmp_gt_f32_vcc", "v_cmp_gt_f32_e32 vcc, {a}, {b}"),
    ("v_cmp_gt_f32_sgpr", "v_cmp_gt_f32_e64 s[30:31], {a}, {b}"),
    ("v_cndmask_b32", "v_cndmask_b32_e32 {d}, {a}, {b}, vcc"),
    ("v_addc_co_u32", "v_addc_co_u32_e32 {d}, vcc, {a}, {b}, vcc"),
    ("v_mov_b32", "v_mov_b32_e32 {d}, {a}"),
    ("v_mov_b64", "v_mov_b64_e32 {d2}, {a2}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {a}, {b}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {a}, {b}, {c}"),
    ("v_min_i16", "v_min_i16 {d}, {a}, {b}"),
    ("v_min_u32_e32", "v_min_u32_e32 {d}, {a}, {b}"),
    ("v_min3_i32", "v_min3_i32 {d}, {a}, {b}, {c}"),
    ("v_min3_u16", "v_min3_u16 {d}, {a}, {b}, {c}"),
    ("v_min_f32_sdwa", "v_min_f32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"),
    ("v_add_u32_sdwa_w1", "v_add_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"),
    ("v_min_f32_dpp", "v_min_f32_dpp {d}, {a}, {b} quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf"),
    ("v_exp_f32", "v_exp_f32_e32 {d}, {a}"),
    ("v_rcp_f32", "v_rcp_f32_e32 {d}, {a}"),
    ("v_permlane32_swap", "v_permlane32_swap_b32_e32 {d}, {a}"),
]

NOPS = 32          # vector operations per loop iteration
# register map of the op kernels (<= 64 VGPRs: 8 waves per SIMD fit)
#   v0, v1    left to the compiler (thread id)     v[2:9]  A / B operands of the MFMAs
#   v[10:25], v[26:41]  two MFMA accumulators (written, never read)
#   v[42:57]  16 source registers                  v[58:63]           6 destinations, used round-robin
SRC0, NSRC, DST0, NDST = 42, 16, 58, 6
MFMA = ["v_mfma_f32_32x32x16_bf16 v[10:25], v[2:5], v[6:9], 0", "v_mfma_f32_32x32x16_bf16 v[26:41], v[2:5], v[6:9], 0"]


def op_line(tmpl, i):
    a = SRC0 + (i % NSRC)
    b = SRC0 + ((i + 1) % NSRC)
    c = SRC0 + ((i + 2) % NSRC)
    d = DST0 + (i % NDST)
    a2 = SRC0 + 2 * (i % (NSRC // 2))
    b2 = SRC0 + 2 * ((i + 1) % (NSRC // 2))
    d2 = DST0 + 2 * (i % (NDST // 2))
    return tmpl.format(d=f"v{d}", a=f"v{a}", b=f"v{b}", c=f"v{c}", d2=f"v[{d2}:{d2 + 1}]", a2=f"v[{a2}:{a2 + 1}]",
                       b2=f"v[{b2}:{b2 + 1}]")


def body_ops(tmpl, with_mfma):
    lines = []
    for i in range(NOPS):
        if with_mfma and i % (NOPS // 2) == 0:
            lines.append(MFMA[i // (NOPS // 2)])
        lines.append(op_line(tmpl, i))
    return lines


# ---- the shipped exact epilogue on fixed registers (vote8x_open / vote8x_close of k4_exact_body.h), two accumulator pairs:
#   pair P: a = v[10+32P : 25+32P], b = v[26+32P : 41+32P];  A/B operands v[2:9];  cnt v74, flg v75, acc v76, dm v77, x0..x3 v78..81,
#   w0 w1 v82 v83, the SAD constant v84  -> 88 VGPRs, 5 waves per SIMD
def epi_open(pa, pb, o):
    a = lambda i: f"v{pa + o + i}"
    b = lambda i: f"v{pb + o + i}"
    return [
        f"v_min3_f32 v78, {a(0)}, {b(0)}, 1.0", f"v_min3_f32 v79, {a(1)}, {b(1)}, 1.0",
        f"v_min3_f32 v80, {a(2)}, {b(2)}, 1.0", f"v_min3_f32 v81, {a(3)}, {b(3)}, 1.0",
        "v_min_f32_e64 v77, |v78|, |v79|", "v_cvt_pknorm_u16_f32 v82, v78, v79",
        f"v_min3_f32 v78, {a(4)}, {b(4)}, 1.0", f"v_min3_f32 v79, {a(5)}, {b(5)}, 1.0",
        "v_min3_f32 v77, v77, |v80|, |v81|", "v_cvt_pknorm_u16_f32 v83, v80, v81",
        f"v_min3_f32 v80, {a(6)}, {b(6)}, 1.0", f"v_min3_f32 v81, {a(7)}, {b(7)}, 1.0",
        "v_add_u32_e32 v76, v82, v83",
        "v_min3_f32 v77, v77, |v78|, |v79|", "v_cvt_pknorm_u16_f32 v82, v78, v79",
        "v_min3_f32 v77, v77, |v80|, |v81|", "v_cvt_pknorm_u16_f32 v83, v80, v81",
        "v_add3_u32 v76, v82, v83, v76",
    ]


def epi_close(pa, pb, o):
    a = lambda i: f"v{pa + o + i}"
    b = lambda i: f"v{pb + o + i}"
    return [
        f"v_min3_f32 v78, {a(0)}, {b(0)}, 1.0", f"v_min3_f32 v79, {a(1)}, {b(1)}, 1.0",
        f"v_min3_f32 v80, {a(2)}, {b(2)}, 1.0", f"v_min3_f32 v81, {a(3)}, {b(3)}, 1.0",
        "v_min3_f32 v77, v77, |v78|, |v79|", "v_cvt_pknorm_u16_f32 v82, v78, v79",
        f"v_min3_f32 v78, {a(4)}, {b(4)}, 1.0", f"v_min3_f32 v79, {a(5)}, {b(5)}, 1.0",
        "v_min3_f32 v77, v77, |v80|, |v81|", "v_cvt_pknorm_u16_f32 v83, v80, v81",
        f"v_min3_f32 v80, {a(6)}, {b(6)}, 1.0", f"v_min3_f32 v81, {a(7)}, {b(7)}, 1.0",
        "v_add3_u32 v76, v82, v83, v76",
        "v_min3_f32 v77, v77, |v78|, |v79|", "v_cvt_pknorm_u16_f32 v82, v78, v79",
        "v_min3_f32 v77, v77, |v80|, |v81|", "v_cvt_pknorm_u16_f32 v83, v80, v81",
        "v_cmp_nle_f32_e32 vcc, 1.0, v77",
        "v_add3_u32 v76, v82, v83, v76",
        "s_nop 0",
        "v_cndmask_b32_e64 v76, v76, 0, vcc",
        "v_addc_co_u32_e32 v75, vcc, v75, v75, vcc",
        "v_add_u32_e32 v74, v74, v76",
    ]


def mfma_pair(p, which):
    base = 10 + 32 * p + 16 * which
    return f"v_mfma_f32_32x32x16_bf16 v[{base}:{base + 15}], v[2:5], v[6:9], 0"


def body_epilogue(mfma, valu):
    """two steps (ping-pong over the two accumulator pairs): step on pair q reads pair q while the MFMAs write pair 1 - q"""
    lines = []
    for q in (0, 1):
        pa, pb = 10 + 32 * q, 26 + 32 * q
        if mfma: lines.append(mfma_pair(1 - q, 0))
        if valu: lines += epi_open(pa, pb, 0)
        if mfma: lines.append(mfma_pair(1 - q, 1))
        if valu: lines += epi_close(pa, pb, 8)
    return lines


# candidate epilogue: i16 norms, band from sum |w| (v_sad_u16 against 0x8000 halves), x by two-input VOP2 min
#   x = min(a, b) [v_min_f32_e32]   w = pknorm_i16(x, x')   T += |w - 0x8000|_u16 (sad)   S += w + w' (add3)
def body_epilogue_sad(mfma, valu):
    lines = []
    for q in (0, 1):
        pa, pb = 10 + 32 * q, 26 + 32 * q
        for hlf in (0, 1):
            if mfma: lines.append(mfma_pair(1 - q, hlf))
            if not valu: continue
            o = 8 * hlf
            a = lambda i: f"v{pa + o + i}"
            b = lambda i: f"v{pb + o + i}"
            lines += [
                f"v_min_f32_e32 v78, {a(0)}, {b(0)}", f"v_min_f32_e32 v79, {a(1)}, {b(1)}",
                f"v_min_f32_e32 v80, {a(2)}, {b(2)}", f"v_min_f32_e32 v81, {a(3)}, {b(3)}",
                "v_cvt_pknorm_i16_f32 v82, v78, v79", "v_cvt_pknorm_i16_f32 v83, v80, v81",
                f"v_min_f32_e32 v78, {a(4)}, {b(4)}", f"v_min_f32_e32 v79, {a(5)}, {b(5)}",
                "v_sad_u16 v77, v82, v84, v77", "v_sad_u16 v77, v83, v84, v77",
                f"v_min_f32_e32 v80, {a(6)}, {b(6)}", f"v_min_f32_e32 v81, {a(7)}, {b(7)}",
                "v_add3_u32 v76, v82, v83, v76",
                "v_cvt_pknorm_i16_f32 v82, v78, v79", "v_cvt_pknorm_i16_f32 v83, v80, v81",
                "v_sad_u16 v77, v82, v84, v77", "v_sad_u16 v77, v83, v84, v77",
                "v_add3_u32 v76, v82, v83, v76",
            ]
    return lines


# candidate (round 4): the MFMAs return dt'' = (s sigma dt + 1) / 2 and cr'' = s sigma cr / 2, so that
#   x = clamp(dt'' - |cr''|)   [v_sub_f32 clamp, FAST class]  is exactly 1.0 (vote) / 0.0 (no vote) outside the band, in (0, 1) inside
#   S += x  [v_add_f32]   Q += x * x  [v_fma_f32]      the cell is clean iff S == Q (x - x^2 > 0 for every x inside (0, 1)), votes = S
# 3 fast-class operations per test + the 4-operation close.  Two chains each for S and Q.
def body_epilogue_float(mfma, valu, chains=2):
    lines = []
    for q in (0, 1):
        pd, pc = 10 + 32 * q, 26 + 32 * q
        d = lambda i: f"v{pd + i}"
        c = lambda i: f"v{pc + i}"
        X = ["v78", "v79", "v80", "v81"]
        SA, SB, QA, QB = "v76", "v82", "v77", "v83"
        for hlf in (0, 1):
            if mfma: lines.append(mfma_pair(1 - q, hlf))
            if not valu: continue
            o = 8 * hlf
            for g in (0, 1):          # groups of four tests
                i0 = o + 4 * g
                lines += [f"v_sub_f32_e64 {X[j]}, {d(i0 + j)}, |{c(i0 + j)}| clamp" for j in range(4)]
                first = hlf == 0 and g == 0
                if first:
                    lines += [f"v_add_f32_e32 {SA}, {X[0]}, {X[1]}", f"v_mul_f32_e32 {QA}, {X[0]}, {X[0]}",
                              f"v_add_f32_e32 {SB}, {X[2]}, {X[3]}", f"v_mul_f32_e32 {QB}, {X[1]}, {X[1]}",
                              f"v_fma_f32 {QA}, {X[2]}, {X[2]}, {QA}", f"v_fma_f32 {QB}, {X[3]}, {X[3]}, {QB}"]
                else:
                    lines += [f"v_add_f32_e32 {SA}, {SA}, {X[0]}", f"v_fma_f32 {QA}, {X[0]}, {X[0]}, {QA}",
                              f"v_add_f32_e32 {SB}, {SB}, {X[1]}", f"v_fma_f32 {QB}, {X[1]}, {X[1]}, {QB}",
                              f"v_add_f32_e32 {SA}, {SA}, {X[2]}", f"v_fma_f32 {QA}, {X[2]}, {X[2]}, {QA}",
                              f"v_add_f32_e32 {SB}, {SB}, {X[3]}", f"v_fma_f32 {QB}, {X[3]}, {X[3]}, {QB}"]
            if hlf == 1:
                lines += [f"v_add_f32_e32 {SA}, {SA}, {SB}", f"v_add_f32_e32 {QA}, {QA}, {QB}",
                          f"v_cmp_neq_f32_e32 vcc, {SA}, {QA}",
                          "s_nop 0",
                          f"v_cndmask_b32_e64 {SA}, {SA}, 0, vcc",
                          "v_addc_co_u32_e32 v75, vcc, v75, v75, vcc",
                          f"v_add_f32_e32 v74, v74, {SA}"]
    return lines


# the same with the band from packed bf16 halves: x (fast), w = cvt_pk_bf16(x, x') , votes += w (pk_add_u16), u = w - 1 (pk_sub_u16),
# mn = min(mn, u) (pk_min_u16): 1 fast + 2 medium-class operations per test
def body_epilogue_pk16(mfma, valu):
    lines = []
    for q in (0, 1):
        pd, pc = 10 + 32 * q, 26 + 32 * q
        d = lambda i: f"v{pd + i}"
        c = lambda i: f"v{pc + i}"
        X = ["v78", "v79", "v80", "v81"]
        for hlf in (0, 1):
            if mfma: lines.append(mfma_pair(1 - q, hlf))
            if not valu: continue
            o = 8 * hlf
            for g in (0, 1):
                i0 = o + 4 * g
                lines += [f"v_sub_f32_e64 {X[j]}, {d(i0 + j)}, |{c(i0 + j)}| clamp" for j in range(4)]
                lines += ["v_cvt_pk_bf16_f32 v82, v78, v79", "v_cvt_pk_bf16_f32 v83, v80, v81",
                          "v_pk_add_u16 v76, v76, v82", "v_pk_sub_u16 v85, v82, v84",
                          "v_pk_add_u16 v76, v76, v83", "v_pk_sub_u16 v86, v83, v84",
                          "v_pk_min_u16 v77, v77, v85", "v_pk_min_u16 v77, v77, v86"]
            if hlf == 1:
                lines += ["v_cmp_gt_u32_e32 vcc, v84, v77", "s_nop 0", "v_cndmask_b32_e64 v76, v76, 0, vcc",
                          "v_addc_co_u32_e32 v75, vcc, v75, v75, vcc", "v_add_u32_e32 v74, v74, v76"]
    return lines


# approximate mode's epilogue for scale: x = clamp(dt - |cr|) (fast), acc = add3(acc, x, x') (slow): 1.5 operations per test
def body_epilogue_approx(mfma, valu):
    lines = []
    for q in (0, 1):
        pd, pc = 10 + 32 * q, 26 + 32 * q
        d = lambda i: f"v{pd + i}"
        c = lambda i: f"v{pc + i}"
        X = ["v78", "v79", "v80", "v81"]
        for hlf in (0, 1):
            if mfma: lines.append(mfma_pair(1 - q, hlf))
            if not valu: continue
            o = 8 * hlf
            for g in (0, 1):
                i0 = o + 4 * g
                lines += [f"v_sub_f32_e64 {X[j]}, {d(i0 + j)}, |{c(i0 + j)}| clamp" for j in range(4)]
                lines += ["v_add3_u32 v74, v78, v79, v74", "v_add3_u32 v74, v80, v81, v74"]
    return lines


# the same with two fast adds instead of one add3
# the shipped round-4 epilogue: x = d - |c| (fast class) instead of min3(a, b, 1); per half step (8 tests): 8 sub + 4 min3 + 4 pknorm +
# 2 add3 (+ 4 to close the cell every second half).  ORDER: where the half's MFMA is issued relative to its operations --
#   0  MFMA, then the subs and the slow operations interleaved (as shipped: hand-placed distances)
#   1  the 8 subs of the half, the MFMA, then the 10 slow operations (they only use the port the MFMA leaves free)
#   2  MFMA, 10 slow operations of the PREVIOUS half's x (kept in 8 more registers), then the subs of this half
def body_epilogue_sub(mfma, valu, order):
    lines = []
    X = ["v78", "v79", "v80", "v81", "v85", "v86", "v87", "v88"]
    def subs(pd, pc, o, regs):
        return [f"v_sub_f32_e64 {regs[j]}, v{pd + o + j}, |v{pc + o + j}|" for j in range(8)]
    def slow(regs, first, close):
        out = [(f"v_min_f32_e64 v77, |{regs[0]}|, |{regs[1]}|" if first else f"v_min3_f32 v77, v77, |{regs[0]}|, |{regs[1]}|"),
               f"v_cvt_pknorm_u16_f32 v82, {regs[0]}, {regs[1]}",
               f"v_min3_f32 v77, v77, |{regs[2]}|, |{regs[3]}|", f"v_cvt_pknorm_u16_f32 v83, {regs[2]}, {regs[3]}",
               ("v_add_u32_e32 v76, v82, v83" if first else "v_add3_u32 v76, v82, v83, v76"),
               f"v_min3_f32 v77, v77, |{regs[4]}|, |{regs[5]}|", f"v_cvt_pknorm_u16_f32 v82, {regs[4]}, {regs[5]}",
               f"v_min3_f32 v77, v77, |{regs[6]}|, |{regs[7]}|", f"v_cvt_pknorm_u16_f32 v83, {regs[6]}, {regs[7]}"]
        if close:
            out += ["v_cmp_nle_f32_e32 vcc, 1.0, v77", "v_add3_u32 v76, v82, v83, v76", "s_nop 0", "v_cndmask_b32_e64 v76, v76, 0, vcc",
                    "v_addc_co_u32_e32 v75, vcc, v75, v75, vcc", "v_add_u32_e32 v74, v74, v76"]
        else:
            out += ["v_add3_u32 v76, v82, v83, v76"]
        return out
    for q in (0, 1):
        pd, pc = 10 + 32 * q, 26 + 32 * q
        for hlf in (0, 1):
            m = [mfma_pair(1 - q, hlf)] if mfma else []
            sb = subs(pd, pc, 8 * hlf, X) if valu else []
            sl = slow(X, hlf == 0, hlf == 1) if valu else []
            if order == 0:      # shipped: MFMA, 4 subs, then interleaved
                if valu:
                    body = sb[:4] + sl[:2] + sb[4:6] + sl[2:4] + sb[6:8] + sl[4:]
                else:
                    body = []
                lines += m + body
            elif order == 1:
                lines += sb + m + sl
            else:
                lines += m + sl + sb   # (the slow operations read the x of the previous half: same registers, timing only)
    return lines

KERNEL = r"""
__global__ __launch_bounds__(256) void {name}(unsigned long long* __restrict__ out, int iters) {{
    extern __shared__ char smem[];
    unsigned long long c0, c1, r0, r1;
    asm volatile(
{init}
        "s_waitcnt lgkmcnt(0)\n"
        "s_memtime %0\n"
        "s_memrealtime %2\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b32 s29, %4\n"
        ".Lloop_%=:\n"
{body}
        "s_sub_u32 s29, s29, 1\n"
        "s_cmp_lg_u32 s29, 0\n"
        "s_cbranch_scc1 .Lloop_%=\n"
        "s_nop 15\n"
        "s_nop 15\n"
        "s_memtime %1\n"
        "s_memrealtime %3\n"
        "s_waitcnt lgkmcnt(0)\n"
        : "=&s"(c0), "=&s"(c1), "=&s"(r0), "=&s"(r1)
        : "s"(iters)
        : {clobbers});
    if ((threadIdx.x & 63) == 0) {{
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
    }}
}}
"""


def kernel(name, lines, nv):
    init = "\n".join(f'        "v_mov_b32 v{i}, 0x3f000000\\n"' for i in range(2, nv))
    body = "\n".join(f'        "{l}\\n"' for l in lines)
    clob = ", ".join([f'"v{i}"' for i in range(2, nv)] + ['"s29"', '"s30"', '"s31"', '"vcc"', '"scc"', '"memory"'])
    return KERNEL.format(name=name, init=init, body=body, clobbers=clob)


MAIN = r"""
struct Case { const char* name; void (*fn)(unsigned long long*, int); int nops; int nmfma; int nv; };
static const Case cases[] = {
%s
};

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : "";
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* dout;
    hipMalloc(&dout, (size_t)cus * 8 * 4 * 2 * 8);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 2);
    printf("# cycles per iteration per SIMD (shader clock, s_memtime), per VALU op where the case has ops; MHz = s_memtime / s_memrealtime(100 MHz)\n");
    printf("# %%-28s %%5s %%5s | waves/SIMD:", "case", "ops", "mfma");
    const int Ws[] = {1, 2, 3, 4, 5, 6, 8};
    for (int W : Ws) printf(" %%13d", W);
    printf("\n");
    for (const Case& c : cases) {
        if (filter[0] && !strstr(c.name, filter)) continue;
        printf("%%-30s %%5d %%5d |            ", c.name, c.nops, c.nmfma);
        double mhz = 0;
        for (int W : Ws) {
            const int maxw = 512 / ((c.nv + 7) / 8 * 8);
            if (W > maxw) { printf(" %%13s", "-"); continue; }
            // W workgroups per CU and no more: each takes 1 / W of 156 KB of the CU's 160 KB of LDS (W + 1 would need > 160 KB for W <= 8)
            const size_t lds = (size_t)(156 * 1024 / W) / 256 * 256;
            hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int occ = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)c.fn, 256, lds);
            if (occ != W) { printf(" occ%%d!=%%d", occ, W); continue; }
            const int iters = 40000 / W;
            hipLaunchKernelGGL(c.fn, dim3(cus * W), dim3(256), lds, 0, dout, iters / 8);  // warm-up
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(c.fn, dim3(cus * W), dim3(256), lds, 0, dout, iters);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
            float wall_ms = 0.f;
            hipEventElapsedTime(&wall_ms, e0, e1);
            hipMemcpy(h.data(), dout, (size_t)cus * W * 4 * 2 * 8, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            const size_t n = (size_t)cus * W * 4;
            for (size_t i = 0; i < n; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
            cyc /= n; rt /= n;
            mhz = cyc / rt * 100.0;
            // per-wave cycles, and (in brackets) the launch's wall time in the same unit: equal when all W workgroups per CU are resident together
            printf(" %%6.1f[%%5.1f]", cyc / ((double)W * iters), wall_ms * 1e-3 * mhz * 1e6 / ((double)W * iters));
        }
        printf("   (%%.0f MHz)\n", mhz);
        fflush(stdout);
    }
    return 0;
}
"""


def main():
    os.makedirs(GEN, exist_ok=True)
    src = ["// GENERATED by tools/ubench_issue.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <stdio.h>",
           "#include <string.h>", "#include <vector>"]
    cases = []
    good_ops = []
    # assemble every op alone first: an unknown mnemonic must not take the whole file down
    for name, tmpl in OPS:
        probe = os.path.join(GEN, "probe.s")
        with open(probe, "w") as f:
            f.write(".text\n" + op_line(tmpl, 0) + "\n" + op_line(tmpl, 5) + "\n")
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950",
                            "-c", probe, "-o", os.path.join(GEN, "probe.o")], capture_output=True, text=True)
        if r.returncode == 0:
            good_ops.append((name, tmpl))
        else:
            print(f"skipped {name}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else 'assembler error'}")
    src.append(kernel("k_mfma_only", [MFMA[0], MFMA[1]], 64))
    cases.append('    {"mfma_only(2)", k_mfma_only, 0, 2, 64},')
    src.append(kernel("k_mfma_only4", [MFMA[0], MFMA[1], MFMA[0], MFMA[1]], 64))
    cases.append('    {"mfma_only(4)", k_mfma_only4, 0, 4, 64},')
    for name, tmpl in good_ops:
        for wm in (0, 1):
            kn = f"k_{name}_{'m' if wm else 'v'}"
            src.append(kernel(kn, body_ops(tmpl, wm), 64))
            cases.append(f'    {{"{name}{" +2mfma" if wm else ""}", {kn}, {NOPS}, {2 * wm}, 64}},')
    def approx_adds(mf, va):
        out = []
        for l in body_epilogue_approx(mf, va):
            if l.startswith("v_add3_u32"):
                _, dst, a, b, _ = l.replace(",", "").split()
                out += [f"v_add_u32_e32 v74, v74, {a}", f"v_add_u32_e32 v76, v76, {b}"]
            else:
                out.append(l)
        return out
    for tag, fn in (("epi_shipped", body_epilogue), ("epi_sad16", body_epilogue_sad), ("epi_float", body_epilogue_float),
                    ("epi_pk16", body_epilogue_pk16), ("epi_approx", body_epilogue_approx), ("epi_approx_add2", approx_adds)):
        for mf, va in ((0, 1), (1, 0), (1, 1)):
            kn = f"k_{tag}_{mf}{va}"
            lines = fn(mf, va)
            nops = sum(1 for l in lines if l.startswith("v_") and "mfma" not in l)
            src.append(kernel(kn, lines, 88))
            cases.append(f'    {{"{tag} 2 steps{" valu" if va else ""}{" +4mfma" if mf else ""}", {kn}, {nops}, {4 * mf}, 88}},')
    for order in (0, 1, 2):
        for mf, va in ((0, 1), (1, 1)):
            kn = f"k_epi_sub{order}_{mf}{va}"
            lines = body_epilogue_sub(mf, va, order)
            nops = sum(1 for l in lines if l.startswith("v_") and "mfma" not in l)
            src.append(kernel(kn, lines, 96))
            cases.append(f'    {{"epi_sub order {order} 2 steps{" valu" if va else ""}{" +4mfma" if mf else ""}", {kn}, {nops}, {4 * mf}, 96}},')
    src.append(MAIN % "\n".join(cases))
    out = os.path.join(GEN, "ubench_issue.hip")
    with open(out, "w") as f:
        f.write("\n".join(src))
    binp = os.path.join(HERE, "ubench_issue.bin")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", out, "-o", binp]
    print(" ".join(cmd))
    subprocess.check_call(cmd)
    print("built", binp)


if __name__ == "__main__":
    sys.exit(main())
