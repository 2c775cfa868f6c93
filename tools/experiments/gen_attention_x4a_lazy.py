#!/usr/bin/env python
"""Generator of det-sam2_amd/csrc/attention_x4a_body.inc: the key loop of the 4-wave / 64-queries-per-wave memory cross-attention
(mode bf16x3k: fp16 single planes) as ONE inline-assembly statement with registers allocated by hand.

Why assembly ("Attention, round 4"): with one wave per SIMD the step is issue-bound, and hipcc neither keeps the
128 registers of Q fragments in the accumulator half nor pipelines the K fragment reads; every C++ form measured slower than the
8-wave kernel.  Here every instruction of the step is placed: per 32-key tile 40 MFMAs (32 cycles each) with <= 14 single-issue
instructions behind each pair.

Register map (one wave = one SIMD, 512 registers):
  a[0:127]    Q^T fragments  qf[qb][ks] = a[(qb*16+ks)*4 ..+3]   (B operand of the score MFMAs)
  a[128:191]  O^T accumulators o[dvb][qb] = a[128+(dvb*2+qb)*16 ..+15]
  v[64:95]    score set A (qb0: 64..79, qb1: 80..95); exponentiated IN PLACE;  v[96:127] score set B
  v[128:143]  P^T as fp16 B fragments pf[qb][st];  v[144:155] K fragment ring (3);  v[156:171] V^T fragments vf[st][dvb]
  v[172:187]  per-lane LDS offsets of the 16 K k-steps;  v[188:191] of the V^T fragments;  v[192:196] LDS-DMA source offsets
  v[197..]    running max / sum, alpha, temporaries
  s[40:63]    addresses, counters (clobbered)
Operands: %[klo] %[khi] %[vlo] %[vhi] %[qlo] %[qhi] %[olo] %[ohi] %[mlo] %[mhi] %[nkt] %[nval] %[ldsb] %[wave] (SGPR), %[lane] (VGPR).
"""
import sys

KT_BYTES, VT_BYTES, V_OFF = 16384, 4096, 65536
SA, SB, PF, KF, VF, KRO, VRO, KDO, VDO = 64, 96, 128, 144, 156, 172, 188, 192, 196
MRUN, LRUN, ALPHA, MNEW, T0 = 197, 199, 201, 203, 205      # (qb0, qb1); T0..: temporaries 205..222
# (v_pk_add_f32 / v_pk_mul_f32 / v_dot2_f32_f16 were measured at 16 / 16 / 10 issue cycles beside MFMAs, v_exp_f32 at 8, plain
#  VALU at 4 - tools/experiments/mfma_issue_cost.hip - so the softmax uses scalar-per-lane instructions only)
import os
ISSUE, GAP = int(os.environ.get('X4A_ISSUE', 4)), int(os.environ.get('X4A_GAP', 26))      # schedule model: issue cycles per filler instruction, filler cycles in the shadow of one MFMA (32 cycles)
TMP = 207
L31, HALF = 223, 224
MASK, H4, NEGINF = 232, 248, 249    # C operand of query block 0's first score MFMA (see CI), 4 * half, -inf
CI = (232, 48)          # per query block 16 registers: -M_ref (+ -inf for keys >= Lk of the last tile) - the first score MFMA's C operand
DELTA = 250             # (qb0, qb1) M_ref now - M_ref the in-flight score set was started with
TAU = 8.0               # re-centre a row only when its tile maximum exceeds M_ref by more than TAU (p <= 2^TAU, exact in the end)
out = []
FLAGS = set(sys.argv[2:])      # timing experiments only (results are wrong): nosoft noexp nokread novread nodma nos nopv nobarrier


def float_hex(x):
    import struct
    return hex(struct.unpack("<I", struct.pack("<f", x))[0])


def e(s):
    op = s.split()[0]
    if "noexp" in FLAGS and op == "v_exp_f32":
        s = s.replace("v_exp_f32", "v_mov_b32")
    if (step.n > 0 or op == "L_loop:") and "strip" in FLAGS and not getattr(e, "epi", False):
        keep = op.startswith("v_mfma") or op.endswith(":") or "s44" in s or op in ("s_branch",) or "L_done" in s
        if "keepwait" in FLAGS and op == "s_waitcnt":
            keep = True
        if "keepsalu" in FLAGS and op.startswith("s_") and "m0" not in s and not op.startswith("s_cbranch") and op not in ("s_barrier", "s_nop", "s_waitcnt"):
            keep = True
        if "keepnop" in FLAGS and op == "s_nop":
            keep = True
        if "keepbr" in FLAGS and (op.startswith("s_cbranch") or op.startswith("s_cmp") or op.startswith("v_cmp") or op == "s_or_b64"):
            keep = True
        if not keep:
            return
    if step.n > 0 and not getattr(e, "epi", False):
        if "loosevm" in FLAGS and s == "s_waitcnt vmcnt(5)":
            s = "s_waitcnt vmcnt(10)"
        if "nowaitk" in FLAGS and s.startswith("s_waitcnt lgkmcnt"):
            return
    if step.n > 0 or op == "L_loop:":      # (inside the loop only)
        if ("nokread" in FLAGS and op == "ds_read_b128" and f"v[{KF}" <= s.split()[1] < f"v[{VF}") or \
           ("novread" in FLAGS and op == "ds_read_b128" and s.split()[1] >= f"v[{VF}") or \
           ("nobarrier" in FLAGS and op == "s_barrier") or ("nodma" in FLAGS and op == "global_load_lds_dwordx4") or \
           ("nos" in FLAGS and op.startswith("v_mfma") and s.split()[1].startswith("v[")) or \
           ("nopv" in FLAGS and op.startswith("v_mfma") and s.split()[1].startswith("a[")):
            return
    out.append(s)


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def qf(qb, ks):
    return ar((qb * 16 + ks) * 4, 4)


def oacc(dvb, qb):
    return ar(128 + (dvb * 2 + qb) * 16, 16)


def kf(i):
    return vr(KF + (i % 3) * 4, 4)


def vf(st, dvb):
    return vr(VF + (st * 2 + dvb) * 4, 4)


def pf(qb, st):
    return vr(PF + (qb * 2 + st) * 4, 4)


def prologue():
    e("s_nop 4")
    # lane constants
    e(f"v_and_b32 {vr(L31)}, 31, %[lane]")
    e(f"v_lshrrev_b32 {vr(HALF)}, 5, %[lane]")
    t0, t1, t2 = TMP, TMP + 1, TMP + 2
    e(f"v_and_b32 {vr(t0)}, 15, {vr(L31)}")                       # ksw = l31 & 15
    e(f"v_lshlrev_b32 {vr(t1)}, 9, {vr(L31)}")                    # l31 * 512
    e(f"v_add_u32 {vr(t1)}, %[ldsb], {vr(t1)}")
    for ks in range(16):
        e(f"v_or_b32 {vr(t2)}, {2 * ks}, {vr(HALF)}")
        e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t0)}")
        e(f"v_lshl_add_u32 {vr(KRO + ks)}, {vr(t2)}, 4, {vr(t1)}")
    for dvb in range(2):
        # row = dvb*32 + l31 ; f = (-(row >> 2)) & 3 ; off = ldsb + V_OFF + row*64 + (((2 st + half) ^ f) << 4)
        e(f"v_add_u32 {vr(t0)}, {dvb * 32}, {vr(L31)}")
        e(f"v_lshrrev_b32 {vr(t1)}, 2, {vr(t0)}")
        e(f"v_sub_u32 {vr(t1)}, 0, {vr(t1)}")
        e(f"v_and_b32 {vr(t1)}, 3, {vr(t1)}")                     # f
        e(f"v_lshlrev_b32 {vr(t0)}, 6, {vr(t0)}")                 # row * 64
        e(f"v_add_u32 {vr(t0)}, %[ldsb], {vr(t0)}")
        e(f"v_add_u32 {vr(t0)}, {V_OFF}, {vr(t0)}")
        for st in range(2):
            e(f"v_or_b32 {vr(t2)}, {2 * st}, {vr(HALF)}")
            e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t1)}")
            e(f"v_lshl_add_u32 {vr(VRO + st * 2 + dvb)}, {vr(t2)}, 4, {vr(t0)}")
    # LDS-DMA source offsets.  K: row0 = 8 wave + half ; koff0 = row0*512 + (((lane & 31) ^ (row0 & 15)) << 4) ; piece j: (koff0 ^ (j << 5)) + 1024 j
    e("s_lshl_b32 s45, %[wave], 3")
    e(f"v_add_u32 {vr(t0)}, s45, {vr(HALF)}")                     # row0
    e(f"v_and_b32 {vr(t1)}, 15, {vr(t0)}")
    e(f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(L31)}")
    e(f"v_lshlrev_b32 {vr(t1)}, 4, {vr(t1)}")
    e(f"v_lshl_add_u32 {vr(KDO)}, {vr(t0)}, 9, {vr(t1)}")
    for j in range(1, 4):
        e(f"v_xor_b32 {vr(t2)}, {j << 5}, {vr(KDO)}")
        e(f"v_add_u32 {vr(KDO + j)}, {j * 1024}, {vr(t2)}")
    # V^T: vrow = 16 wave + (lane >> 2) ; f = (-(vrow >> 2)) & 3 ; off = vrow*64 + (((lane & 3) ^ f) << 4)
    e("s_lshl_b32 s45, %[wave], 4")
    e(f"v_lshrrev_b32 {vr(t0)}, 2, %[lane]")
    e(f"v_add_u32 {vr(t0)}, s45, {vr(t0)}")
    e(f"v_lshrrev_b32 {vr(t1)}, 2, {vr(t0)}")
    e(f"v_sub_u32 {vr(t1)}, 0, {vr(t1)}")
    e(f"v_and_b32 {vr(t1)}, 3, {vr(t1)}")
    e(f"v_and_b32 {vr(t2)}, 3, %[lane]")
    e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t1)}")
    e(f"v_lshlrev_b32 {vr(t2)}, 4, {vr(t2)}")
    e(f"v_lshl_add_u32 {vr(VDO)}, {vr(t0)}, 6, {vr(t2)}")
    # LDS destinations of this wave's pieces: K base s48 = ldsb + wave*4096 ; V base s49 = ldsb + V_OFF + wave*1024
    e("s_lshl_b32 s45, %[wave], 12")
    e("s_add_u32 s48, %[ldsb], s45")
    e("s_lshl_b32 s45, %[wave], 10")
    e("s_add_u32 s49, %[ldsb], s45")
    e(f"s_add_u32 s49, s49, {V_OFF}")
    e("s_sub_u32 s46, %[nkt], 1")                                 # last tile
    e("s_mov_b32 s44, 0")                                         # t
    # score mask: zero until the last tile's scores are started (step)
    for r in range(16):
        e(f"v_mov_b32 {vr(CI[0] + r)}, 0")
        e(f"v_mov_b32 {vr(CI[1] + r)}, 0")
    e(f"v_mov_b32 {vr(DELTA)}, 0")
    e(f"v_mov_b32 {vr(DELTA + 1)}, 0")
    e(f"s_mov_b32 s56, {float_hex(TAU)}")
    e(f"v_lshlrev_b32 {vr(H4)}, 2, {vr(HALF)}")
    e(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
    # running statistics
    for qb in range(2):
        e(f"v_mov_b32 {vr(MRUN + qb)}, 0")        # M_ref: 0 until the first tile re-centres it (forced, threshold -inf)
        e(f"v_mov_b32 {vr(LRUN + qb)}, 0")
    for i in range(128, 192):
        e(f"v_accvgpr_write_b32 a{i}, 0")
    # Q fragments: 32 x 16 bytes per lane from [frag][lane][16 B]
    e(f"v_lshlrev_b32 {vr(t0)}, 4, %[lane]")
    e("s_mov_b32 s50, %[qlo]")
    e("s_mov_b32 s51, %[qhi]")
    for i in range(32):
        e(f"global_load_dwordx4 {ar(i * 4, 4)}, {vr(t0)}, s[50:51] offset:{(i % 4) * 1024}")
        if i % 4 == 3 and i != 31:
            e("s_add_u32 s50, s50, 4096")
            e("s_addc_u32 s51, s51, 0")
    # first tiles: K(0) V(0) K(1) V(1) K(2)
    for kt, slot, with_v in ((0, 0, True), (1, 1, True), (2, 2, False)):
        for ins in dma_k(kt_imm=kt, slot=slot) + (dma_v(kt_imm=kt, slot=slot) if with_v else []):
            e(ins)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # scores of tile 0 -> set A
    e(f"ds_read_b128 {kf(0)}, {vr(KRO + 0)}")
    e(f"ds_read_b128 {kf(1)}, {vr(KRO + 1)} ")
    for ks in range(16):
        if ks + 2 < 16:
            e(f"ds_read_b128 {kf(ks + 2)}, {vr(KRO + ks + 2)}")
        e(f"s_waitcnt lgkmcnt({min(2, 15 - ks)})")
        for qb in range(2):
            c = "0" if ks == 0 else vr(SA + qb * 16, 16)
            e(f"v_mfma_f32_32x32x16_f16 {vr(SA + qb * 16, 16)}, {kf(ks)}, {qf(qb, ks)}, {c}")
    e("s_nop 15")
    e("s_nop 7")


def dma_k(slot, kt_imm=None, t_plus=None):
    """copy K tile min(t + t_plus, nkt - 1) (or the constant tile kt_imm, clamped) into ring slot `slot` -> instruction list"""
    L = []
    if kt_imm is not None:
        L.append(f"s_min_u32 s45, {kt_imm}, s46")
    else:
        L.append(f"s_add_u32 s45, s44, {t_plus}")
        L.append("s_min_u32 s45, s45, s46")
    L.append("s_lshl_b32 s45, s45, 14")
    L.append("s_add_u32 s40, %[klo], s45")
    L.append("s_addc_u32 s41, %[khi], 0")
    for j in range(4):
        L.append(f"s_add_u32 m0, s48, {slot * KT_BYTES + j * 1024}")
        L.append("s_nop 0")
        L.append(f"global_load_lds_dwordx4 {vr(KDO + j)}, s[40:41]")
    return L


def dma_v(slot, kt_imm=None, t_plus=None):
    L = []
    if kt_imm is not None:
        L.append(f"s_min_u32 s45, {kt_imm}, s46")
    else:
        L.append(f"s_add_u32 s45, s44, {t_plus}")
        L.append("s_min_u32 s45, s45, s46")
    L.append("s_lshl_b32 s45, s45, 12")
    L.append("s_add_u32 s42, %[vlo], s45")
    L.append("s_addc_u32 s43, %[vhi], 0")
    L.append(f"s_add_u32 m0, s49, {slot * VT_BYTES}")
    L.append("s_nop 0")
    L.append(f"global_load_lds_dwordx4 {vr(VDO)}, s[42:43]")
    return L


def cost(ins):
    """issue cycles a filler takes from the one wave of its SIMD (model: one instruction per ISSUE cycles, transcendentals 16)"""
    op = ins.split()[0]
    if op.startswith("v_exp"):
        return 8
    if op == "s_nop":
        return int(ins.split()[1]) + 1
    return ISSUE


def softmax_stream(cur, n):
    """the online softmax of the scores in set `cur` (in place) as an ordered filler list of (instruction | block, tag)"""
    S = lambda qb, i: vr(cur + qb * 16 + i)   # noqa: E731
    F = []
    add = lambda ins, tag=None: F.append((ins, tag))   # noqa: E731
    # maxima of the 16 scores of each query block -> T0 + qb
    for qb in range(2):
        add(f"v_max3_f32 {vr(T0 + qb)}, {S(qb, 0)}, {S(qb, 1)}, {S(qb, 2)}")
        for i in range(3, 15, 2):
            add(f"v_max3_f32 {vr(T0 + qb)}, {vr(T0 + qb)}, {S(qb, i)}, {S(qb, i + 1)}")
        add(f"v_max_f32 {vr(T0 + qb)}, {vr(T0 + qb)}, {S(qb, 15)}")
    # across the lane halves -> the tile maximum RELATIVE to the reference M_ref the set was started with (the first score MFMA's C
    # operand carries -M_ref, so the scores arrive already shifted: no subtraction per score); T = that maximum relative to the
    # CURRENT M_ref; the row is re-centred only when T > threshold: d = T, else 0; sh = what has to come off this set's scores
    for qb in range(2):
        add(f"v_mov_b32 {vr(TMP + qb)}, {vr(T0 + qb)}")
    add("s_nop 1")
    for qb in range(2):
        add(f"v_permlane32_swap_b32 {vr(T0 + qb)}, {vr(TMP + qb)}")
    add("s_nop 1")
    for qb in range(2):
        add(f"v_max_f32 {vr(T0 + qb)}, {vr(T0 + qb)}, {vr(TMP + qb)}")
    for qb in range(2):
        add(f"v_sub_f32 {vr(TMP + 2 + qb)}, {vr(T0 + qb)}, {vr(DELTA + qb)}")
    for qb in range(2):
        add(f"v_cmp_lt_f32 vcc, s47, {vr(TMP + 2 + qb)}")
        add(f"v_cndmask_b32 {vr(MNEW + qb)}, 0, {vr(TMP + 2 + qb)}, vcc")          # d
    for qb in range(2):
        add(f"v_add_f32 {vr(TMP + 2 + qb)}, {vr(DELTA + qb)}, {vr(MNEW + qb)}")   # sh
    # one block, taken when a row of the wave is re-centred or the set was started before the last re-centring
    lab = f"L_norescale_{n}"
    blk = [f"v_cmp_neq_f32 vcc, 0, {vr(TMP + 2)}", f"v_cmp_neq_f32 s[52:53], 0, {vr(TMP + 3)}", "s_or_b64 vcc, vcc, s[52:53]",
           "s_nop 1", f"s_cbranch_vccz {lab}"]
    for qb in range(2):
        for i in range(16):
            blk.append(f"v_sub_f32 {S(qb, i)}, {S(qb, i)}, {vr(TMP + 2 + qb)}")
    for qb in range(2):                                   # alpha = 2^-max(d, 0) (d < 0 only on the first tile, where O = l = 0)
        blk.append(f"v_sub_f32 {vr(TMP + 4 + qb)}, 0, {vr(MNEW + qb)}")
        blk.append(f"v_min_f32 {vr(TMP + 4 + qb)}, 0, {vr(TMP + 4 + qb)}")
    for qb in range(2):
        blk.append(f"v_exp_f32 {vr(ALPHA + qb)}, {vr(TMP + 4 + qb)}")
    for qb in range(2):
        blk.append(f"v_add_f32 {vr(MRUN + qb)}, {vr(MRUN + qb)}, {vr(MNEW + qb)}")
        blk.append(f"v_mov_b32 {vr(DELTA + qb)}, {vr(MNEW + qb)}")
        for r in range(16):
            blk.append(f"v_sub_f32 {vr(CI[qb] + r)}, {vr(CI[qb] + r)}, {vr(MNEW + qb)}")
    for qb in range(2):
        blk.append(f"v_mul_f32 {vr(LRUN + qb)}, {vr(LRUN + qb)}, {vr(ALPHA + qb)}")
    for dvb in range(2):
        for qb in range(2):
            base = 128 + (dvb * 2 + qb) * 16
            for r in range(0, 16, 4):
                for i in range(4):
                    blk.append(f"v_accvgpr_read_b32 {vr(TMP + 4 + i)}, a{base + r + i}")
                for i in range(4):
                    blk.append(f"v_mul_f32 {vr(TMP + 4 + i)}, {vr(TMP + 4 + i)}, {vr(ALPHA + qb)}")
                for i in range(4):
                    blk.append(f"v_accvgpr_write_b32 a{base + r + i}, {vr(TMP + 4 + i)}")
    blk += ["s_nop 7", f"{lab}:"]
    F.append((blk, None))
    # p = exp2(s' ) in place (s' = s - M_ref), pair by pair; one pair behind: the fp16 packing
    pairs = [(qb, 4 * st + w) for st in range(2) for qb in range(2) for w in range(4)]
    prev = None

    def finish(pr):
        qb, j = pr
        st, w = j // 4, j % 4
        add(f"v_cvt_pk_f16_f32 {vr(PF + (qb * 2 + st) * 4 + w)}, {S(qb, 2 * j)}, {S(qb, 2 * j + 1)}", ("pf", qb, st) if w == 3 else None)

    for qb, j in pairs:
        add(f"v_exp_f32 {S(qb, 2 * j)}, {S(qb, 2 * j)}")
        add(f"v_exp_f32 {S(qb, 2 * j + 1)}, {S(qb, 2 * j + 1)}")
        if prev is not None:
            finish(prev)
        prev = (qb, j)
    add("s_nop 0")
    finish(prev)
    # row sums: l += sum(p) (a re-centring has scaled l in its block)
    for qb in range(2):
        t = TMP + 8 + qb * 4
        P = lambda i, qb=qb: S(qb, i)   # noqa: E731
        for ins in [f"v_add_f32 {vr(t)}, {P(0)}, {P(1)}", f"v_add_f32 {vr(t + 1)}, {P(2)}, {P(3)}",
                    f"v_add_f32 {vr(t + 2)}, {P(4)}, {P(5)}", f"v_add_f32 {vr(t + 3)}, {P(6)}, {P(7)}",
                    f"v_add_f32 {vr(t)}, {vr(t)}, {P(8)}", f"v_add_f32 {vr(t + 1)}, {vr(t + 1)}, {P(9)}",
                    f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {P(10)}", f"v_add_f32 {vr(t + 3)}, {vr(t + 3)}, {P(11)}",
                    f"v_add_f32 {vr(t)}, {vr(t)}, {P(12)}", f"v_add_f32 {vr(t + 1)}, {vr(t + 1)}, {P(13)}",
                    f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {P(14)}", f"v_add_f32 {vr(t + 3)}, {vr(t + 3)}, {P(15)}",
                    f"v_add_f32 {vr(t)}, {vr(t)}, {vr(t + 1)}", f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {vr(t + 3)}",
                    f"v_add_f32 {vr(t)}, {vr(t)}, {vr(t + 2)}",
                    f"v_add_f32 {vr(LRUN + qb)}, {vr(LRUN + qb)}, {vr(t)}"]:
            add(ins)
    if "nosoft" in FLAGS:
        F = [("s_nop 0", tag) for _, tag in F if tag]
    return F


def step(sl, cur, nxt):
    kslot, vslot = (sl + 1) & 3, sl & 3
    n = step.n
    step.n += 1
    # the scores started in this step are those of tile t + 1: when it is the last one, keys >= nval (of its 32) get -inf through
    # the first MFMA's C operand.  Register r of a lane in half h holds key (r & 3) + 8 (r >> 2) + 4 h.
    e("s_cmp_eq_u32 s44, 0")                          # re-centring threshold: -inf for the first tile (always), TAU after
    e("s_cselect_b32 s47, 0xff800000, s56")
    lab = f"L_nomask_{n}"
    e("s_add_u32 s45, s44, 2")
    e("s_cmp_eq_u32 s45, %[nkt]")
    e(f"s_cbranch_scc0 {lab}")
    for r in range(16):
        e(f"s_sub_i32 s45, %[nval], {(r & 3) + 8 * (r >> 2)}")
        e(f"v_cmp_le_i32 vcc, s45, {vr(H4)}")
        for qb in range(2):
            e(f"v_cndmask_b32 {vr(CI[qb] + r)}, {vr(CI[qb] + r)}, {vr(NEGINF)}, vcc")
    e("s_nop 4")
    e(f"{lab}:")
    # LDS reads first issued: the four V^T fragments of tile t, K fragments 0 and 1 of tile t + 1
    for st in range(2):
        for dvb in range(2):
            e(f"ds_read_b128 {vf(st, dvb)}, {vr(VRO + st * 2 + dvb)} offset:{vslot * VT_BYTES}")
    e(f"ds_read_b128 {kf(0)}, {vr(KRO + 0)} offset:{kslot * KT_BYTES}")
    e(f"ds_read_b128 {kf(1)}, {vr(KRO + 1)} offset:{kslot * KT_BYTES}")
    # fillers, in order: copies of the tiles three / two steps ahead, then the softmax of the current scores
    F = [(i, None) for i in dma_k(slot=(sl + 3) & 3, t_plus=3) + dma_v(slot=(sl + 2) & 3, t_plus=2)] + softmax_stream(cur, n)
    pos = [0]
    done = set()
    debt = [0.0]

    def emit_one():
        ins, tag = F[pos[0]]
        pos[0] += 1
        c = 0
        for line in (ins if isinstance(ins, list) else [ins]):
            e(line)
            c += cost(line) if isinstance(ins, str) else 0
        if isinstance(ins, list):
            c = 5 * ISSUE            # the block's usual path: the branch over it
        if tag:
            done.add(tag)
        return c

    def fill(budget):
        """fillers worth `budget` issue cycles in the shadow of the MFMA just issued (over / undershoot carried)"""
        debt[0] += budget
        while pos[0] < len(F) and debt[0] > 0:
            debt[0] -= emit_one()

    def flush_until(tag):
        any_ = False
        while tag not in done:
            debt[0] -= emit_one()
            any_ = True
        if any_:
            e("s_nop 1")           # VALU write of a P fragment -> MFMA read

    for ks in range(16):
        if ks + 2 < 16:
            e(f"ds_read_b128 {kf(ks + 2)}, {vr(KRO + ks + 2)} offset:{kslot * KT_BYTES}")
        e(f"s_waitcnt lgkmcnt({min(2, 15 - ks)})")
        for qb in range(2):
            c = vr(CI[qb], 16) if ks == 0 else vr(nxt + qb * 16, 16)
            e(f"v_mfma_f32_32x32x16_f16 {vr(nxt + qb * 16, 16)}, {kf(ks)}, {qf(qb, ks)}, {c}")
            fill(GAP - (2 * ISSUE if qb == 1 and ks + 1 < 16 else 0))     # (the next slot's read + wait are issued in this gap)
    # P.V
    for st in range(2):
        for dvb in range(2):
            for qb in range(2):
                flush_until(("pf", qb, st))
                e(f"v_mfma_f32_32x32x16_f16 {oacc(dvb, qb)}, {vf(st, dvb)}, {pf(qb, st)}, {oacc(dvb, qb)}")
                fill(GAP)
    while pos[0] < len(F):
        emit_one()
    e("s_waitcnt vmcnt(5)")
    e("s_barrier")
    e("s_add_u32 s44, s44, 1")
    e("s_cmp_lt_u32 s44, %[nkt]")
    e("s_cbranch_scc0 L_done")


step.n = 0


def epilogue():
    e.epi = True
    e("L_done:")
    e("s_waitcnt vmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")
    # l over the lane halves; (m, l) and the unnormalised O^T rows to the scratch of k_w8_merge (nsplit = 1)
    t0, t1, t2 = TMP, TMP + 1, TMP + 2
    for qb in range(2):
        e(f"v_mov_b32 {vr(t0 + qb)}, {vr(LRUN + qb)}")
    e("s_nop 1")
    for qb in range(2):
        e(f"v_permlane32_swap_b32 {vr(LRUN + qb)}, {vr(t0 + qb)}")
    e("s_nop 1")
    for qb in range(2):
        e(f"v_add_f32 {vr(LRUN + qb)}, {vr(LRUN + qb)}, {vr(t0 + qb)}")
    # row (of this wave's 64) = qb*32 + l31 ; O offset = row*256 + half*16 bytes ; ml offset = row*8
    e(f"v_lshlrev_b32 {vr(t2)}, 8, {vr(L31)}")
    e(f"v_lshl_add_u32 {vr(t2)}, {vr(HALF)}, 4, {vr(t2)}")
    e(f"v_lshlrev_b32 {vr(t2 + 1)}, 3, {vr(L31)}")
    e("s_mov_b32 s50, %[olo]")
    e("s_mov_b32 s51, %[ohi]")
    e("s_mov_b32 s54, %[mlo]")
    e("s_mov_b32 s55, %[mhi]")
    e("s_nop 4")
    for qb in range(2):
        if qb == 1:
            e(f"v_add_u32 {vr(t2)}, {32 * 256}, {vr(t2)}")
            e(f"v_add_u32 {vr(t2 + 1)}, {32 * 8}, {vr(t2 + 1)}")
        for dvb in range(2):
            base = 128 + (dvb * 2 + qb) * 16
            for g in range(4):
                e(f"global_store_dwordx4 {vr(t2)}, {ar(base + 4 * g, 4)}, s[50:51] offset:{(32 * dvb + 8 * g) * 4}")
        ml = TMP + 9 + 2 * qb          # an even-aligned pair per query block
        e(f"v_mov_b32 {vr(ml)}, {vr(MRUN + qb)}")
        e(f"v_mov_b32 {vr(ml + 1)}, {vr(LRUN + qb)}")
        e("s_nop 1")
        e(f"global_store_dwordx2 {vr(t2 + 1)}, {vr(ml, 2)}, s[54:55]")
        e("s_nop 1")
    e("s_waitcnt vmcnt(0)")


def main():
    prologue()
    e("L_loop:")
    step(0, SA, SB)
    step(1, SB, SA)
    step(2, SA, SB)
    step(3, SB, SA)
    e("s_branch L_loop")
    epilogue()
    body = "\n".join('    "' + ln + '\\n\\t"' for ln in out)
    clob = [f'"v{i}"' for i in range(48, 252)] + [f'"a{i}"' for i in range(0, 192)] + [f'"s{i}"' for i in range(40, 57)] + ['"vcc"', '"scc"', '"memory"']
    txt = ("// GENERATED by tools/gen/gen_attention_x4a.py - do not edit.\n"
           f"// {len(out)} instructions\n"
           "#define X4A_ASM_BODY \\\n" + " \\\n".join('    "' + ln + '\\n\\t"' for ln in out) + "\n"
           "#define X4A_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    open(sys.argv[1], "w").write(txt)
    print(len(out), "instructions")


if __name__ == "__main__":
    main()
