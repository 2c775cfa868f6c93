#!/usr/bin/env python
"""Generator of det-sam2_amd/csrc/attention_x4s_body.inc: the key loop of the memory SELF-attention of mode bf16x3k (D = DV = 256,
fp16 single planes) in the form of the assembly cross-attention (tools/gen/gen_attention_x4a.py): 4 waves,
ONE per SIMD, here 32 queries per wave - the 256-wide value rows need 128 accumulator registers per 32 queries - so a wave reads
every K / V^T fragment once per 32 queries where the 8-wave kernel (16 queries per wave, 16 waves per CU) reads it once per 16.

Register map (one wave = one SIMD):
  a[0:63]     Q^T fragments  qf[ks] = a[4 ks ..+3]            (B operand of the score MFMAs)
  a[64:191]   O^T accumulators o[dvb] = a[64 + 16 dvb ..+15]  (dvb = 0..7)
  v[64:79]    score set A, v[80:95] score set B (exponentiated IN PLACE)
  v[96:103]   P^T as fp16 B fragments pf[st];  v[104:115] K fragment ring (3);  v[116:179] V^T fragments vf[st][dvb]
  v[180:195]  per-lane LDS offsets of the 16 K k-steps;  v[196:197] of the V^T fragments (per st; dvb by immediate offset)
  v[198:201]  LDS-DMA source offsets of this wave's 4 K pieces, v[202:205] of its 4 V^T pieces
  v[206..]    running max / sum, alpha, temporaries;  v[232:247] additive mask of the last key tile
  s[40:56]    addresses, counters (clobbered)
Operands: as the cross-attention's.
"""
import os
import sys

KT_BYTES, VT_BYTES, V_OFF, NDVB = 16384, 16384, 65536, 8
SA, SB, PF, KF, VF, KRO, VRO, KDO, VDO = 64, 80, 96, 104, 116, 180, 196, 198, 202
MRUN, LRUN, ALPHA, MNEW, T0 = 206, 207, 208, 209, 210
TMP = 211                      # 211..226
L31, HALF = 227, 228
MASK, H4, NEGINF = 232, 248, 249
ISSUE, GAP = int(os.environ.get("X4S_ISSUE", 4)), int(os.environ.get("X4S_GAP", 26))
out = []


def e(s):
    out.append(s)


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def qf(ks):
    return ar(ks * 4, 4)


def oacc(dvb):
    return ar(64 + dvb * 16, 16)


def kf(i):
    return vr(KF + (i % 3) * 4, 4)


def vf(st, dvb):
    return vr(VF + (st * NDVB + dvb) * 4, 4)


def pf(st):
    return vr(PF + st * 4, 4)


def dma_k(slot, kt_imm=None, t_plus=None):
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
    L.append("s_lshl_b32 s45, s45, 14")                      # V^T tile: 256 rows x 64 bytes
    L.append("s_add_u32 s42, %[vlo], s45")
    L.append("s_addc_u32 s43, %[vhi], 0")
    for j in range(4):
        L.append(f"s_add_u32 m0, s49, {slot * VT_BYTES + j * 1024}")
        L.append("s_nop 0")
        L.append(f"global_load_lds_dwordx4 {vr(VDO + j)}, s[42:43]")
    return L


def prologue():
    e("s_nop 4")
    e(f"v_and_b32 {vr(L31)}, 31, %[lane]")
    e(f"v_lshrrev_b32 {vr(HALF)}, 5, %[lane]")
    t0, t1, t2 = TMP, TMP + 1, TMP + 2
    # K fragment reads: row l31 of the tile (512 bytes), chunk (2 ks + half) ^ (l31 & 15)
    e(f"v_and_b32 {vr(t0)}, 15, {vr(L31)}")
    e(f"v_lshlrev_b32 {vr(t1)}, 9, {vr(L31)}")
    e(f"v_add_u32 {vr(t1)}, %[ldsb], {vr(t1)}")
    for ks in range(16):
        e(f"v_or_b32 {vr(t2)}, {2 * ks}, {vr(HALF)}")
        e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t0)}")
        e(f"v_lshl_add_u32 {vr(KRO + ks)}, {vr(t2)}, 4, {vr(t1)}")
    # V^T fragment reads: row dvb * 32 + l31 (64 bytes; dvb by immediate offset), chunk (2 st + half) ^ f, f = (-(l31 >> 2)) & 3
    e(f"v_lshrrev_b32 {vr(t1)}, 2, {vr(L31)}")
    e(f"v_sub_u32 {vr(t1)}, 0, {vr(t1)}")
    e(f"v_and_b32 {vr(t1)}, 3, {vr(t1)}")
    e(f"v_lshlrev_b32 {vr(t0)}, 6, {vr(L31)}")
    e(f"v_add_u32 {vr(t0)}, %[ldsb], {vr(t0)}")
    e(f"v_add_u32 {vr(t0)}, {V_OFF}, {vr(t0)}")
    for st in range(2):
        e(f"v_or_b32 {vr(t2)}, {2 * st}, {vr(HALF)}")
        e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t1)}")
        e(f"v_lshl_add_u32 {vr(VRO + st)}, {vr(t2)}, 4, {vr(t0)}")
    # LDS-DMA sources.  K: row0 = 8 wave + half ; koff0 = row0 * 512 + (((lane & 31) ^ (row0 & 15)) << 4) ; piece j: (koff0 ^ (j << 5)) + 1024 j
    e("s_lshl_b32 s45, %[wave], 3")
    e(f"v_add_u32 {vr(t0)}, s45, {vr(HALF)}")
    e(f"v_and_b32 {vr(t1)}, 15, {vr(t0)}")
    e(f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(L31)}")
    e(f"v_lshlrev_b32 {vr(t1)}, 4, {vr(t1)}")
    e(f"v_lshl_add_u32 {vr(KDO)}, {vr(t0)}, 9, {vr(t1)}")
    for j in range(1, 4):
        e(f"v_xor_b32 {vr(t2)}, {j << 5}, {vr(KDO)}")
        e(f"v_add_u32 {vr(KDO + j)}, {j * 1024}, {vr(t2)}")
    # V^T: piece p = 4 wave + j holds rows 16 p .. 16 p + 15 ; lane -> row 16 p + (lane >> 2), chunk (lane & 3) ^ f,
    # f = (-(row >> 2)) & 3 = (-(lane >> 4)) & 3
    e("s_lshl_b32 s45, %[wave], 6")
    e(f"v_lshrrev_b32 {vr(t0)}, 2, %[lane]")
    e(f"v_add_u32 {vr(t0)}, s45, {vr(t0)}")                    # row of piece 0
    e(f"v_lshrrev_b32 {vr(t1)}, 4, %[lane]")
    e(f"v_sub_u32 {vr(t1)}, 0, {vr(t1)}")
    e(f"v_and_b32 {vr(t1)}, 3, {vr(t1)}")
    e(f"v_and_b32 {vr(t2)}, 3, %[lane]")
    e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t1)}")
    e(f"v_lshlrev_b32 {vr(t2)}, 4, {vr(t2)}")
    e(f"v_lshl_add_u32 {vr(VDO)}, {vr(t0)}, 6, {vr(t2)}")
    for j in range(1, 4):
        e(f"v_add_u32 {vr(VDO + j)}, {j * 1024}, {vr(VDO)}")
    # LDS destinations of this wave's pieces: K base s48 = ldsb + wave * 4096 ; V base s49 = ldsb + V_OFF + wave * 4096
    e("s_lshl_b32 s45, %[wave], 12")
    e("s_add_u32 s48, %[ldsb], s45")
    e(f"s_add_u32 s49, s48, {V_OFF}")
    e("s_sub_u32 s46, %[nkt], 1")
    e("s_mov_b32 s44, 0")
    for r in range(16):
        e(f"v_mov_b32 {vr(MASK + r)}, 0")
    e(f"v_lshlrev_b32 {vr(H4)}, 2, {vr(HALF)}")
    e(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
    e(f"v_mov_b32 {vr(MRUN)}, 0xff800000")
    e(f"v_mov_b32 {vr(LRUN)}, 0")
    for i in range(64, 192):
        e(f"v_accvgpr_write_b32 a{i}, 0")
    # Q fragments: 16 x 16 bytes per lane from [frag][lane][16 B]
    e(f"v_lshlrev_b32 {vr(t0)}, 4, %[lane]")
    e("s_mov_b32 s50, %[qlo]")
    e("s_mov_b32 s51, %[qhi]")
    for i in range(16):
        e(f"global_load_dwordx4 {ar(i * 4, 4)}, {vr(t0)}, s[50:51] offset:{(i % 4) * 1024}")
        if i % 4 == 3 and i != 15:
            e("s_add_u32 s50, s50, 4096")
            e("s_addc_u32 s51, s51, 0")
    for kt, slot, with_v in ((0, 0, True), (1, 1, True), (2, 2, False)):
        for ins in dma_k(kt_imm=kt, slot=slot) + (dma_v(kt_imm=kt, slot=slot) if with_v else []):
            e(ins)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # scores of tile 0 -> set A
    e(f"ds_read_b128 {kf(0)}, {vr(KRO + 0)}")
    e(f"ds_read_b128 {kf(1)}, {vr(KRO + 1)}")
    for ks in range(16):
        if ks + 2 < 16:
            e(f"ds_read_b128 {kf(ks + 2)}, {vr(KRO + ks + 2)}")
        e(f"s_waitcnt lgkmcnt({min(2, 15 - ks)})")
        c = "0" if ks == 0 else vr(SA, 16)
        e(f"v_mfma_f32_32x32x16_f16 {vr(SA, 16)}, {kf(ks)}, {qf(ks)}, {c}")
    e("s_nop 15")
    e("s_nop 7")


def cost(ins):
    op = ins.split()[0]
    if op.startswith("v_exp"):
        return 8
    if op == "s_nop":
        return int(ins.split()[1]) + 1
    return ISSUE


def softmax_stream(cur, n):
    S = lambda i: vr(cur + i)   # noqa: E731
    F = []
    add = lambda ins, tag=None: F.append((ins, tag))   # noqa: E731
    add(f"v_max3_f32 {vr(T0)}, {S(0)}, {S(1)}, {S(2)}")
    for i in range(3, 15, 2):
        add(f"v_max3_f32 {vr(T0)}, {vr(T0)}, {S(i)}, {S(i + 1)}")
    add(f"v_max_f32 {vr(T0)}, {vr(T0)}, {S(15)}")
    add(f"v_mov_b32 {vr(TMP)}, {vr(T0)}")
    add("s_nop 1")
    add(f"v_permlane32_swap_b32 {vr(T0)}, {vr(TMP)}")
    add("s_nop 1")
    add(f"v_max3_f32 {vr(MNEW)}, {vr(T0)}, {vr(TMP)}, {vr(MRUN)}")
    add(f"v_sub_f32 {vr(TMP + 2)}, {vr(MRUN)}, {vr(MNEW)}")
    add(f"v_exp_f32 {vr(ALPHA)}, {vr(TMP + 2)}")
    add(f"v_mov_b32 {vr(MRUN)}, {vr(MNEW)}")
    lab = f"L_norescale_{n}"
    blk = [f"v_cmp_neq_f32 vcc, 1.0, {vr(ALPHA)}", "s_nop 1", f"s_cbranch_vccz {lab}"]
    for dvb in range(NDVB):
        base = 64 + dvb * 16
        for r in range(0, 16, 4):
            for i in range(4):
                blk.append(f"v_accvgpr_read_b32 {vr(TMP + 4 + i)}, a{base + r + i}")
            for i in range(4):
                blk.append(f"v_mul_f32 {vr(TMP + 4 + i)}, {vr(TMP + 4 + i)}, {vr(ALPHA)}")
            for i in range(4):
                blk.append(f"v_accvgpr_write_b32 a{base + r + i}, {vr(TMP + 4 + i)}")
    blk += ["s_nop 7", f"{lab}:"]
    F.append((blk, None))
    prev = None

    def finish(j):
        st, w = j // 4, j % 4
        add(f"v_cvt_pk_f16_f32 {vr(PF + st * 4 + w)}, {S(2 * j)}, {S(2 * j + 1)}", ("pf", st) if w == 3 else None)

    for j in range(8):
        add(f"v_sub_f32 {S(2 * j)}, {S(2 * j)}, {vr(MNEW)}")
        add(f"v_sub_f32 {S(2 * j + 1)}, {S(2 * j + 1)}, {vr(MNEW)}")
        add(f"v_exp_f32 {S(2 * j)}, {S(2 * j)}")
        add(f"v_exp_f32 {S(2 * j + 1)}, {S(2 * j + 1)}")
        if prev is not None:
            finish(prev)
        prev = j
    add("s_nop 0")
    finish(prev)
    t = TMP + 8
    for ins in [f"v_add_f32 {vr(t)}, {S(0)}, {S(1)}", f"v_add_f32 {vr(t + 1)}, {S(2)}, {S(3)}",
                f"v_add_f32 {vr(t + 2)}, {S(4)}, {S(5)}", f"v_add_f32 {vr(t + 3)}, {S(6)}, {S(7)}",
                f"v_add_f32 {vr(t)}, {vr(t)}, {S(8)}", f"v_add_f32 {vr(t + 1)}, {vr(t + 1)}, {S(9)}",
                f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {S(10)}", f"v_add_f32 {vr(t + 3)}, {vr(t + 3)}, {S(11)}",
                f"v_add_f32 {vr(t)}, {vr(t)}, {S(12)}", f"v_add_f32 {vr(t + 1)}, {vr(t + 1)}, {S(13)}",
                f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {S(14)}", f"v_add_f32 {vr(t + 3)}, {vr(t + 3)}, {S(15)}",
                f"v_add_f32 {vr(t)}, {vr(t)}, {vr(t + 1)}", f"v_add_f32 {vr(t + 2)}, {vr(t + 2)}, {vr(t + 3)}",
                f"v_add_f32 {vr(t)}, {vr(t)}, {vr(t + 2)}",
                f"v_fma_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(ALPHA)}, {vr(t)}"]:
        add(ins)
    return F


def step(sl, cur, nxt):
    kslot, vslot = (sl + 1) & 3, sl & 3
    n = step.n
    step.n += 1
    lab = f"L_nomask_{n}"
    e("s_add_u32 s45, s44, 2")
    e("s_cmp_eq_u32 s45, %[nkt]")
    e(f"s_cbranch_scc0 {lab}")
    for r in range(16):
        e(f"s_sub_i32 s45, %[nval], {(r & 3) + 8 * (r >> 2)}")
        e(f"v_cmp_le_i32 vcc, s45, {vr(H4)}")
        e(f"v_cndmask_b32 {vr(MASK + r)}, 0, {vr(NEGINF)}, vcc")
    e("s_nop 4")
    e(f"{lab}:")
    for st in range(2):
        for dvb in range(NDVB):
            e(f"ds_read_b128 {vf(st, dvb)}, {vr(VRO + st)} offset:{vslot * VT_BYTES + dvb * 2048}")
    e(f"ds_read_b128 {kf(0)}, {vr(KRO + 0)} offset:{kslot * KT_BYTES}")
    e(f"ds_read_b128 {kf(1)}, {vr(KRO + 1)} offset:{kslot * KT_BYTES}")
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
            c = 3 * ISSUE
        if tag:
            done.add(tag)
        return c

    def fill(budget):
        debt[0] += budget
        while pos[0] < len(F) and debt[0] > 0:
            debt[0] -= emit_one()

    def flush_until(tag):
        any_ = False
        while tag not in done:
            debt[0] -= emit_one()
            any_ = True
        if any_:
            e("s_nop 1")

    for ks in range(16):
        if ks + 2 < 16:
            e(f"ds_read_b128 {kf(ks + 2)}, {vr(KRO + ks + 2)} offset:{kslot * KT_BYTES}")
        e(f"s_waitcnt lgkmcnt({min(2, 15 - ks)})")
        c = vr(MASK, 16) if ks == 0 else vr(nxt, 16)
        e(f"v_mfma_f32_32x32x16_f16 {vr(nxt, 16)}, {kf(ks)}, {qf(ks)}, {c}")
        fill(GAP - (2 * ISSUE if ks + 1 < 16 else 0))
    for st in range(2):
        for dvb in range(NDVB):
            flush_until(("pf", st))
            e(f"v_mfma_f32_32x32x16_f16 {oacc(dvb)}, {vf(st, dvb)}, {pf(st)}, {oacc(dvb)}")
            fill(GAP)
    while pos[0] < len(F):
        emit_one()
    e("s_waitcnt vmcnt(8)")
    e("s_barrier")
    e("s_add_u32 s44, s44, 1")
    e("s_cmp_lt_u32 s44, %[nkt]")
    e("s_cbranch_scc0 L_done")


step.n = 0


def epilogue():
    e("L_done:")
    e("s_waitcnt vmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")
    t0, t2 = TMP, TMP + 2
    e(f"v_mov_b32 {vr(t0)}, {vr(LRUN)}")
    e("s_nop 1")
    e(f"v_permlane32_swap_b32 {vr(LRUN)}, {vr(t0)}")
    e("s_nop 1")
    e(f"v_add_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(t0)}")
    # row (of this wave's 32) = l31 ; O offset = row * 1024 + half * 16 bytes ; ml offset = row * 8
    e(f"v_lshlrev_b32 {vr(t2)}, 10, {vr(L31)}")
    e(f"v_lshl_add_u32 {vr(t2)}, {vr(HALF)}, 4, {vr(t2)}")
    e(f"v_lshlrev_b32 {vr(t2 + 1)}, 3, {vr(L31)}")
    e("s_mov_b32 s50, %[olo]")
    e("s_mov_b32 s51, %[ohi]")
    e("s_mov_b32 s54, %[mlo]")
    e("s_mov_b32 s55, %[mhi]")
    e("s_nop 4")
    for dvb in range(NDVB):
        for g in range(4):
            e(f"global_store_dwordx4 {vr(t2)}, {ar(64 + dvb * 16 + 4 * g, 4)}, s[50:51] offset:{(32 * dvb + 8 * g) * 4}")
    ml = TMP + 11                  # (an even-aligned pair)
    e(f"v_mov_b32 {vr(ml)}, {vr(MRUN)}")
    e(f"v_mov_b32 {vr(ml + 1)}, {vr(LRUN)}")
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
    clob = [f'"v{i}"' for i in range(64, 250)] + [f'"a{i}"' for i in range(0, 192)] + [f'"s{i}"' for i in range(40, 56)] + ['"vcc"', '"scc"', '"memory"']
    txt = ("// GENERATED by tools/gen/gen_attention_x4s.py - do not edit.  The key loop of k_attention_x4s as one asm statement.\n"
           "#define X4S_ASM_BODY \\\n" + body.replace("\n", " \\\n") + "\n\n#define X4S_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    open(sys.argv[1], "w").write(txt)
    print(len(out), "instructions")


main()
