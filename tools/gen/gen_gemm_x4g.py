#!/usr/bin/env python
"""Generator of det-sam2_amd/csrc/gemm_x4g_body_<cfg>_<epi>.inc: the persistent bf16x3 GEMM whose EPILOGUE OF TILE i RUNS UNDER THE
MAIN LOOP OF TILE i+1, as ONE inline-assembly statement with registers allocated by hand.

Why (DESIGN.md "GEMM findings" 8): in k_gemm_split_pp256 the GELU / split / store epilogue of a 256x256 tile is as
long as its 18-K-tile main loop and the matrix pipe idles through it; the 8-wave kernels are at 228 of 256 registers, so a second
accumulator set does not fit.  Here: 4 waves, ONE per SIMD, 512 registers each.

  wave tile   (32 MBW) x (32 NBW) blocks of 32x32; workgroup tile (64 MBW) x (64 NBW): cfg 42 -> 256x128, cfg 23 -> 128x192
  a[0:16 NB)      live accumulators of tile i+1 (NB = MBW NBW blocks, v_mfma_f32_32x32x16_bf16, 3 terms per product, the order of
                  every other tile kernel: a_lo w_hi, a_hi w_lo, a_hi w_hi per 16-deep sub-step -> bit-identical results)
  a[128:128+16 NB) PARKED accumulators of tile i: copied there (v_accvgpr_mov) block by block during the last sub-step of tile i and
                  drained - bias, GELU, bf16 split, lane-pair exchange, stores - as fillers behind the MFMAs of the first 16 K tiles
                  of tile i+1 (one of 16 drain steps per K tile; K >= 17 * 32)
  LDS             3-stage ring of 32-deep K tiles {A hi, A lo, W hi, W lo}, rows of 64 B, 16-byte chunks XOR-swizzled by
                  (row >> 2) & 3 (the pp256 image), filled by LDS-DMA two K tiles ahead; ONE s_barrier per K tile placed after 3/4 of
                  its MFMAs; the fragments of the next K tile's first sub-step are read behind the last quarter
  epilogue forms  e1: C = acc + bias (fp32);  e2: planes(gelu(acc + bias)) (bf16 hi / lo, pairs of columns exchanged between
                  neighbouring lanes so that a store writes 64-byte row segments);  e3: C = acc + bias + R (fp32, C may alias R)

Instruction stream per wave: prologue, then K-tile bodies {D0..D15 (with drain step d), plain, last (parks)} and a tail (the drain of
the workgroup's last tile, not overlapped).  Every wait is COUNTED: vmcnt by simulating the in-order VMEM stream (LDS-DMA, stores,
residual / bias loads), lgkmcnt by simulating the LDS stream.  `lint()` checks the manual hazards (trans -> use, VALU vcc / SGPR ->
mask, DPP source, m0 -> LDS-DMA, readfirstlane -> SALU).

MX form (cfg "23m", round 6; det-sam2_amd/csrc/common.h "MX" operand planes): the two-MFMA-equivalent product a w ~= a16 w16 + a8 wl8 + al8 w8.
Plane 1 of an operand is fp16, plane 2 one 16-bit word per element (value byte | remainder byte, swapped for weights) with plane 1's
layout, so DMA, LDS image, swizzle and the four chunk addresses of a K tile are those of the bf16 form.  Per K tile (64 deep) and
32 x 32 block: four v_mfma_f32_32x32x16_f16 (one per 16-deep sub-step, fragment sets X / Y alternate) and TWO
v_mfma_scale_f32_32x32x64_f8f6f4 whose 8-register operands are the plane-2 chunk pairs (q' = 0, 1) and (2, 3) - each instruction is both
cross terms of 32 k's, scaled by the E8M0 byte 127 + EA - LW on the A side (static scales; the K order inside a tile is free because
every block carries the same scale; operand layout and scale semantics measured by tools/ubench/mx_*.hip).  Body: fp16 sub-steps 0..2
(behind them: the fp16 fragments of the next sub-step AND the 20 plane-2 reads of THIS tile), barrier, sub-step 3, the 12 fp8 MFMAs
(64 cycles each: the DMA block of K tile t+2 and the drain fillers are paced under them).  e2 writes MX ACTIVATION planes
(v_cvt_pk_f16_f32, v_cvt_scalef32_pk_fp8_f32 x 2, v_perm_b32 under MODE.FP16_OVFL = saturating conversions).

usage: gen_gemm_x4g.py <out.inc> <cfg: 42|23|23m> <epi: e1|e2|e3> [flags: nodrain nostore nodma ...]  (flags: timing experiments only)
"""
import os
import sys

# ------------------------------------------------------------------------------------------------ configuration
OUT, CFG, EPI = sys.argv[1], sys.argv[2], sys.argv[3]
FLAGS = set(sys.argv[4:])
MX = CFG.endswith("m")          # "23m": fp16 hi.hi + ONE scaled fp8 MFMA per 32 k for both cross terms (module docstring, "MX form")
MBW, NBW = int(CFG[0]), int(CFG[1])
NB = MBW * NBW
TM, TN = 64 * MBW, 64 * NBW
KSUB = int(os.environ.get("X4G_KSUB", 2 if CFG in ("23", "23m") else 1))   # 32-deep sub-tiles per LDS stage: the 128 x 192 tile stages 64-deep K
RBYTES = 64 * KSUB                              # tiles, so that an LDS-DMA piece is 8 rows x 128 B = FULL L2 lines (profiles/r05_l_dma.txt:
PA_B, PW_B = TM * RBYTES, TN * RBYTES             # the skeleton of the loop halves its time against 16 rows x 64 B); bytes of one plane of a stage
STAGE = 2 * (PA_B + PW_B)
if KSUB == 2:
    NSTAGE, LOOK, DMA_POST = 2, 2, True       # two 80 KiB stages; the DMA of K tile t+2 is issued BEHIND the barrier of body t (which frees its stage)
else:
    NSTAGE = int(os.environ.get("X4G_NSTAGE", min(4, 160 * 1024 // STAGE)))   # 3 stages of 48 KiB (cfg 42)
    LOOK, DMA_POST = NSTAGE - 1, False        # K tiles in flight ahead of the one being multiplied; DMA issued at the top of a body
NQ = 2 * KSUB                                 # 16-deep sub-steps of a body (fragment sets X, Y alternate)
NSLOT = (NQ + 2) * NB if MX else 3 * NB * NQ   # MFMAs per body (MX: NQ fp16 sub-steps + 2 scaled fp8 MFMAs per block)
BAR_SLOT = (NQ - 1) * NB if MX else (NSLOT * 3) // 4
NDRAIN = 16 // KSUB                           # bodies that carry a drain step
RP = MBW // 2                                 # (narrow drain) row pairs (of 8 per block) per step, 16 steps
STEPS_PER_MB = 8 // RP
PA_N, PW_N = MBW * KSUB, NBW * KSUB           # LDS-DMA pieces (1 KiB) per plane, wave and body
NP = 0 if 'nodma' in FLAGS else 2 * ((0 if 'onlyw' in FLAGS else PA_N) + (0 if 'onlya' in FLAGS else PW_N))   # pieces per wave and body
ISSUE = int(os.environ.get("X4G_ISSUE", 4))
# form of the drain per epilogue, measured in one call (profiles/r05_f_stores.txt; X4G_WIDE / X4G_STMOD override for A/B builds):
#   e1 (fp32, 453 MB at the qkv shape): quad transposes + 16-byte non-temporal stores 384 us, plain 403, dword stores 440 (pp256: 405)
#   e2 (GELU + planes): the lane-pair exchange + dword stores 521 us, transposes + 8-byte stores 555 (the extra VALU work costs more)
#   e3 (residual): dword loads / stores 176 - 178 us, transposes 186
_DEF_WIDE, _DEF_STMOD = {"e1": ("1", "nt"), "e2": ("0", ""), "e3": ("0", "nt")}[EPI]
_stmod = os.environ.get("X4G_STMOD", _DEF_STMOD)
STMOD = (" " + _stmod) if _stmod and _stmod != "none" else ""      # cache policy of the result stores (nt | sc1 | sc0 sc1 | none)
STDEFER = os.environ.get("X4G_STDEFER", "0") == "1"
GAP = int(os.environ.get("X4G_GAP", 24))      # filler issue cycles hidden behind one MFMA (32 cycles)
GAP8 = int(os.environ.get("X4G_GAP8", 56))    # ... behind one v_mfma_scale_f32_32x32x64_f8f6f4 (16 passes = 64 cycles)
# MX: 1 = two barriers per K tile, the two operand planes of a stage re-filled separately (body_mx2).  Measured (profiles/r06_ab_mx_split.txt):
# residual form -9 %, GELU form -3 %, +0.5 % frames/s - and, as any change of the accumulation order does, other boundary pixels flip: the
# worst 1 - IoU of the measured-shape fixtures read 4.4e-4 instead of 2.9e-4 (same product error against fp64, 8.0e-6).  The committed
# bodies are the one-barrier form (0): half a percent is not worth the thinner-looking margin; the form stays here for a 3-stage future.
MX_SPLIT = int(os.environ.get("X4G_MX_SPLIT", 0))
DMA_PACE = int(os.environ.get("X4G_DMA_PACE", 1))   # MX: 1 = the DMA block paced under the MFMAs behind the barrier, 0 = one burst
RD_EARLY = int(os.environ.get("X4G_RD_EARLY", 0))   # MX: the last LDS read of a stage is issued this many MFMA slots before its barrier
assert not MX or (KSUB == 2 and CFG == "23m"), "the MX form exists for the 128 x 192 tile with 64-deep K tiles"

# VGPR map (v0..v31 are left to the compiler)
if not MX:
    FX, FY = 32, 88                           # fragment sets: ah[mb] +4mb, al[mb] +16+4mb, wh[nb] +32+4nb, wl[nb] +44+4nb
    RA, RW = 144, 148                         # LDS read bases of the current stage, one per sub-step (<= 4 each)
    DOA, DOW = 152, 156                       # LDS-DMA lane offsets: A pieces (<= 4), W pieces (<= 6)
    VOC, VOR, VOP, SEL = 164, 165, 166, 167   # lane offsets of C / R / plane stores, v_perm selector of the lane-pair exchange
    BIAS, KC2, L31, HALF = 168, 171, 172, 173
    T0, NTMP = 176, 40                        # temporaries 176..215
    RBUF = 216                                # e3: two sets of residual values, loaded one drain step ahead
else:
    # MX form: fp16 fragment sets of ONE 16-deep sub-step (ah[mb] +4mb, wh[nb] +8+4nb: 20 registers), and the fp8 operands of the WHOLE
    # K tile (a8[mb][j] / w8[nb][j], 8 registers each: j = the 32 k's of chunks 2j', 2j'+1 of plane 2)
    FX, FY = 32, 52
    F8A, F8W = 72, 72 + 16 * MBW              # 72..103, 104..151
    RA, RW = 152, 156
    DOA, DOW = 160, 164
    VOC, VOR, VOP, SEL = 172, 173, 174, 175
    BIAS, KC2, L31, HALF = 176, 179, 180, 181
    SCA, SCB = 182, 183                       # E8M0 scale bytes of the fp8 operands (byte 0 of each)
    T0, NTMP = 184, 40                        # temporaries 184..223
    RBUF = 224
RS = 12 if KSUB == 2 else 8                   # registers of a set
# SGPR map (s32 / s33 and s96.. are reserved by the compiler; s0..s15 are left to it)
S_AH, S_AL, S_WH, S_WL, S_BIAS, S_C, S_R, S_CH, S_CL = 36, 38, 40, 42, 44, 46, 48, 50, 52
S_LDA, S_LDW, S_LDC, S_LDR, S_LDCP, S_NK = 54, 55, 56, 57, 58, 59
P_AH, P_AL, P_WH, P_WL = 60, 62, 64, 66       # DMA cursor: source bases of its K tile
S_KD, S_DDST, S_KC, S_TLEFT, S_RDELTA, S_NKM1, S_LEND, S_CST = 68, 69, 70, 71, 72, 73, 74, 75
W_C, W_R, W_P, W_B = 76, 77, 78, 79           # this wave's offsets inside a tile (bytes)
CUR, NXT = 80, 84                             # tile entries {c_off, r_off, p_off, b_off} of the compute / DMA tile
D_C, D_R, D_H, D_L = 88, 90, 92, 94           # drain bases (64-bit) of the parked tile (s90..s95 double as prologue temporaries)
ROWA, ROWB, ROWC, ROWD = 16, 18, 20, 22       # row address pairs
ST = 24                                       # scalar temporaries 24..29
G_C0, G_C1 = 30, 31                           # GELU constants
S_STG, S_NSTG2 = 34, 35                       # STAGE, -(NSTAGE - 1) STAGE
S_TLI = 29                                    # index of the DMA cursor's tile in the workgroup's tile list (ST + 5)
S_M1, S_M2 = 8, 10                            # lane masks (lane & 1), (lane & 2) of the quad transposes
S_WDA, S_WDW, S_DSTA, S_DSTW = 12, 13, 14, 15  # this wave's row offset inside a stage's A / W planes; DMA destinations of this K tile
# MX e2 (planes of gelu(acc + bias) as MX activation planes): the fp32 drain bases D_C / D_R are not used by this form - their
# registers hold the conversion constants (2^EA and 2^-LA as f32: v_cvt_scalef32_pk_fp8_f32 DIVIDES by its scale operand - measured,
# tools/ubench/mx_cvt_probe.hip; the byte interleave selector of v_perm_b32)
S_SCV, S_SCR, S_PSEL = 88, 89, 90

# MX form: static power-of-two scales of the fp8 operands (det-sam2_amd/csrc/kernels.h DS2_MX_*; gemm_x4g.hip static_asserts the match).
# plane 2 of an activation element = e4m3(a 2^-EA) | e4m3((a - fp16(a)) 2^LA) << 8, of a weight element = e4m3((w - fp16(w)) 2^LW) |
# e4m3(w 2^-EW) << 8; EA + LA = EW + LW makes both cross terms of a byte pair carry the same factor 2^(EA - LW).
MX_EA, MX_LA, MX_EW, MX_LW = -2, 14, -6, 18
assert MX_EA + MX_LA == MX_EW + MX_LW
MX_SCALE_A = 127 + MX_EA - MX_LW

out = []
lint_off = [False]


def e(s):
    out.append(s)


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


def frag(base, kind, i):
    if MX:
        return vr(base + {"ah": 0, "wh": 4 * MBW}[kind] + 4 * i, 4)
    return vr(base + {"ah": 0, "al": 16, "wh": 32, "wl": 44}[kind] + 4 * i, 4)


def f8(kind, i, j, part=None):
    """fp8 operand j (0 | 1) of row block i: 8 registers; part 0 | 1 = its first / second ds_read_b128"""
    b = (F8A if kind == "a" else F8W) + (2 * i + j) * 8
    return vr(b, 8) if part is None else vr(b + 4 * part, 4)


def acc(b):
    return ar(16 * b, 16)


def cost(ins):
    op = ins.split()[0]
    if op in ("v_exp_f32", "v_rcp_f32"):
        return 8
    if op == "s_nop":
        return int(ins.split()[1]) + 1
    if op.endswith(":"):
        return 0
    return ISSUE


# ------------------------------------------------------------------------------------------------ counted waits
class Stream:
    """in-order completion stream (vmcnt / lgkmcnt): ops are appended by name; need(name) returns the s_waitcnt count that guarantees
    completion of the LAST op of that name (None if already known complete)"""

    def __init__(self, carry=()):
        self.ops = list(carry)
        self.done = 0

    def issue(self, name):
        self.ops.append(name)

    def need(self, name):
        idx = max(i for i, n in enumerate(self.ops) if n == name)
        if idx < self.done:
            return None
        self.done = idx + 1
        return len(self.ops) - idx - 1

    def wait_all(self):
        self.done = len(self.ops)


# canonical LDS sequence issued behind the barrier of every body (and by the prologue): the tile-list entry, then the fragments of
# the next K tile's sub-step 0 in the order its MFMAs want them
def x_order():
    o = []
    if MX:          # fp16 fragments of a sub-step in the order its MFMAs (block (0,0), (0,1), ...) want them
        return [("ah", 0)] + [("wh", nb) for nb in range(NBW)] + [("ah", mb) for mb in range(1, MBW)]
    for mb in range(MBW):
        o.append(("al", mb))
        if mb == 0:
            o += [("wh", nb) for nb in range(NBW)]
    o += [("ah", mb) for mb in range(MBW)] + [("wl", nb) for nb in range(NBW)]
    return o


def y_order():
    return x_order()


POST_BAR = [f"X{k}{i}" for k, i in x_order()]


def f8_order():
    """plane-2 reads of a K tile: (kind, row block, q') with q' = 0..3 the chunk pair index (the SAME four addresses as the fp16
    sub-steps); operand j = q' // 2 first for every block, so that the first scaled MFMAs can start"""
    o = []
    for j in range(2):
        for kind, n in (("a", MBW), ("w", NBW)):
            for i in range(n):
                o += [(kind, i, 2 * j), (kind, i, 2 * j + 1)]
    return o


def read8_ins(kind, i, qp):
    base = (RA if kind == "a" else RW) + qp
    off = i * 32 * RBYTES + (PA_B if kind == "a" else PW_B)
    return f"ds_read_b128 {f8(kind, i, qp // 2, qp % 2)}, {vr(base)} offset:{off}"


def read_ins(setbase, kind, i, q):
    base = (RA if kind[0] == "a" else RW) + q
    plane = 1 if kind[1] == "l" else 0
    off = i * 32 * RBYTES + plane * (PA_B if kind[0] == "a" else PW_B)
    return f"ds_read_b128 {frag(setbase, kind, i)}, {vr(base)} offset:{off}"


# ------------------------------------------------------------------------------------------------ pieces of a body
def dma_pieces(planes=(0, 1)):
    """LDS-DMA of the DMA cursor's K tile into stage S_DDST: NP (m0 write, copy) pairs (planes: which operand planes)"""
    L = []
    ldmod = " nt" if "ldnt" in FLAGS else ""
    for plane, (pa, pw) in enumerate(((P_AH, P_WH), (P_AL, P_WL))):
        if plane not in planes:
            continue
        for j in range(PA_N):
            if "onlyw" in FLAGS:
                continue
            if "areg" in FLAGS:
                L.append(["s_nop 0", f"global_load_dwordx4 {vr(232 + 4 * ((plane * PA_N + j) % 6), 4)}, {vr(DOA + j)}, {sr(pa, 2)}"])
                continue
            L.append([f"s_add_u32 m0, {sr(S_DSTA)}, {plane * PA_B + j * 1024}", f"global_load_lds_dwordx4 {vr(DOA + j)}, {sr(pa, 2)}{ldmod}"])
        for j in range(PW_N):
            if "onlya" in FLAGS:
                continue
            if "wreg" in FLAGS:          # (timing experiment: the W pieces as plain loads into registers instead of LDS-DMA)
                L.append(["s_nop 0", f"global_load_dwordx4 {vr(232 + 4 * ((plane * PW_N + j) % 6), 4)}, {vr(DOW + j)}, {sr(pw, 2)}"])
                continue
            if "areg" in FLAGS:
                pass
            L.append([f"s_add_u32 m0, {sr(S_DSTW)}, {2 * PA_B + plane * PW_B + j * 1024}", f"global_load_lds_dwordx4 {vr(DOW + j)}, {sr(pw, 2)}"])
    return L


def dma_dst_setup():
    """destinations of this wave's pieces in stage S_DDST (wave w stages rows [16 MBW w, +16 MBW) of A and [16 NBW w, ..) of W)"""
    return [f"s_add_u32 {sr(S_DSTA)}, {sr(S_DDST)}, {sr(S_WDA)}", f"s_add_u32 {sr(S_DSTW)}, {sr(S_DDST)}, {sr(S_WDW)}"]


def tile_entry(dst):
    """tile S_TLI of this workgroup's list -> DMA source bases + entry offsets {c, r, p, b} into s[dst..dst+3].  The list lives in TWO
    VGPRs (entry i in lane i & 63 of %[tl0] / %[tl1]: tile_m << 16 | tile_n, written by the kernel's C++ prologue; entries past
    the last tile repeat it - the DMA cursor runs ahead of the end)."""
    t0, t1, t2 = ST, ST + 1, ST + 4
    L = [f"s_and_b32 {sr(t2)}, {sr(S_TLI)}, {0 if 'sametile' in FLAGS else 63}", "s_nop 3",
         f"v_readlane_b32 {sr(t0)}, %[tl0], {sr(t2)}", f"v_readlane_b32 {sr(t1)}, %[tl1], {sr(t2)}", "s_nop 4",
         f"s_cmp_lt_u32 {sr(S_TLI)}, 64", f"s_cselect_b32 {sr(t0)}, {sr(t0)}, {sr(t1)}",
         f"s_lshr_b32 {sr(t1)}, {sr(t0)}, 16", f"s_and_b32 {sr(t0)}, {sr(t0)}, 0xffff",           # t1 = tile_m, t0 = tile_n
         f"s_mul_i32 {sr(t1)}, {sr(t1)}, {TM}", f"s_mul_i32 {sr(t0)}, {sr(t0)}, {TN}",             # m0, n0
         # A / W planes: byte offsets of the tile's first rows
         f"s_mul_i32 {sr(t2)}, {sr(t1)}, {sr(S_LDA)}", f"s_lshl_b32 {sr(t2)}, {sr(t2)}, 1",
         f"s_add_u32 {sr(P_AH)}, {sr(S_AH)}, {sr(t2)}", f"s_addc_u32 {sr(P_AH + 1)}, {sr(S_AH + 1)}, 0",
         f"s_add_u32 {sr(P_AL)}, {sr(S_AL)}, {sr(t2)}", f"s_addc_u32 {sr(P_AL + 1)}, {sr(S_AL + 1)}, 0",
         f"s_mul_i32 {sr(t2)}, {sr(t0)}, {sr(S_LDW)}", f"s_lshl_b32 {sr(t2)}, {sr(t2)}, 1",
         f"s_add_u32 {sr(P_WH)}, {sr(S_WH)}, {sr(t2)}", f"s_addc_u32 {sr(P_WH + 1)}, {sr(S_WH + 1)}, 0",
         f"s_add_u32 {sr(P_WL)}, {sr(S_WL)}, {sr(t2)}", f"s_addc_u32 {sr(P_WL + 1)}, {sr(S_WL + 1)}, 0"]
    for k, (ld, sh) in enumerate(((S_LDC, 2), (S_LDR, 2), (S_LDCP, 1))):                             # (m0 ld + n0) * element size
        L += [f"s_mul_i32 {sr(t2)}, {sr(t1)}, {sr(ld)}", f"s_add_u32 {sr(t2)}, {sr(t2)}, {sr(t0)}", f"s_lshl_b32 {sr(dst + k)}, {sr(t2)}, {sh}"]
    L.append(f"s_lshl_b32 {sr(dst + 3)}, {sr(t0)}, 2")
    return L


def advance_block(n):
    """DMA cursor -> next K tile (wrap: K tile 0 of the next tile of the list)"""
    L = [f"s_add_u32 {sr(S_KD)}, {sr(S_KD)}, 1", f"s_cmp_lt_u32 {sr(S_KD)}, {sr(S_NK)}", f"s_cbranch_scc1 L_adv_{n}",
         f"s_mov_b32 {sr(S_KD)}, 0", f"s_add_u32 {sr(S_TLI)}, {sr(S_TLI)}, 1"]
    L += tile_entry(NXT)
    L += [f"s_branch L_advd_{n}", f"L_adv_{n}:"]
    for p in (P_AH, P_AL, P_WH, P_WL):
        L += [f"s_add_u32 {sr(p)}, {sr(p)}, {128 if 'p128' in FLAGS else 64 * KSUB}", f"s_addc_u32 {sr(p + 1)}, {sr(p + 1)}, 0"]
    L.append(f"L_advd_{n}:")
    return L


def rotate_dma_dst():
    return [f"s_add_u32 {sr(S_DDST)}, {sr(S_DDST)}, {STAGE}", f"s_cmp_eq_u32 {sr(S_DDST)}, {sr(S_LEND)}",
            f"s_cselect_b32 {sr(S_DDST)}, %[ldsb], {sr(S_DDST)}"]


def rotate_read_delta():
    return [f"s_add_u32 {sr(S_CST)}, {sr(S_CST)}, 1", f"s_cmp_eq_u32 {sr(S_CST)}, {NSTAGE}", f"s_cselect_b32 {sr(S_CST)}, 0, {sr(S_CST)}",
            f"s_cselect_b32 {sr(S_RDELTA)}, {sr(S_NSTG2)}, {sr(S_STG)}"]


# ------------------------------------------------------------------------------------------------ drain steps
# The parked tile is drained in UNITS of 4 accumulator registers (mb, g, nb): registers 4g..4g+3 of block (mb, nb) = rows
# 32 mb + 8 g + 4 half + {0,1,2,3}, column 32 nb + l31 of the wave tile.  A 4 x 4 transpose inside every quad of lanes (two DPP
# butterfly stages) turns "lane = column, registers = rows" into "lane i of a quad = row i, registers = 4 consecutive columns", so
# that the result leaves in 16-byte (fp32) / 8-byte (bf16 planes) pieces and a store instruction writes 8 rows x 128 / 64 bytes:
# the kernel is bound by the number of VMEM instructions (LDS-DMA + stores share the CU's one address path), not by their bytes.
UNITS = [(mb, g, nb) for mb in range(MBW) for g in range(4) for nb in range(NBW)]


def step_units(d):
    return UNITS[d * len(UNITS) // NDRAIN:(d + 1) * len(UNITS) // NDRAIN]


def park(mb, nb, r):
    return ar(128 + (mb * NBW + nb) * 16 + r)


XOR1 = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
XOR2 = "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"


class Drain:
    """the filler stream of drain step d of the parked tile.  Items: instruction strings; ("vm", ins, name) VMEM ops; ("needprev", d)"""

    def __init__(self, d):
        self.F = []
        self.tix = 0
        self.d = d
        getattr(self, EPI)(step_units(d))

    def tmp(self, n):
        if self.tix + n > NTMP:
            self.tix = 0
        b = T0 + self.tix
        self.tix += n
        return b

    def add(self, s):
        self.F.append(s)

    def rowaddr(self, pair, base, row, ld, esz):
        self.add(f"s_mul_i32 {sr(ST + 4)}, {sr(ld)}, {row * esz}")
        self.add(f"s_add_u32 {sr(pair)}, {sr(base)}, {sr(ST + 4)}")
        self.add(f"s_addc_u32 {sr(pair + 1)}, {sr(base + 1)}, 0")

    def store(self, width, voff, data, pair, off):
        if "nostore" in FLAGS:
            return
        op = {4: "global_store_dwordx4", 2: "global_store_dwordx2"}[width]
        self.F.append(("vm", f"{op} {vr(voff)}, {vr(data, width)}, {sr(pair, 2)} offset:{off}{STMOD}", "st"))

    def transpose4(self, t):
        """4 x 4 transpose of v[t..t+3] across the lanes of every quad (x: two scratch registers)"""
        A = self.add
        x = self.tmp(4)
        for (mask, ctl, pairs) in ((S_M2, XOR2, ((0, 2), (1, 3))), (S_M1, XOR1, ((0, 1), (2, 3)))):
            for j, (a, b) in enumerate(pairs):       # lanes with the bit set send their a, the others their b
                A(f"v_cndmask_b32_e64 {vr(x + j)}, {vr(t + b)}, {vr(t + a)}, {sr(mask, 2)}")
            A("s_nop 0")
            for j, (a, b) in enumerate(pairs):
                A(f"v_mov_b32_dpp {vr(x + 2 + j)}, {vr(x + j)} {ctl}")
            for j, (a, b) in enumerate(pairs):
                A(f"v_cndmask_b32_e64 {vr(t + a)}, {vr(t + a)}, {vr(x + 2 + j)}, {sr(mask, 2)}")
                A(f"v_cndmask_b32_e64 {vr(t + b)}, {vr(x + 2 + j)}, {vr(t + b)}, {sr(mask, 2)}")

    def read_bias(self, t, mb, g, nb):
        for k in range(4):
            self.add(f"v_accvgpr_read_b32 {vr(t + k)}, {park(mb, nb, 4 * g + k)}")
        for k in range(4):
            self.add(f"v_add_f32 {vr(t + k)}, {vr(t + k)}, {vr(BIAS + nb)}")

    def e1(self, units):
        for i, (mb, g, nb) in enumerate(units):
            pair = (ROWA, ROWB)[i % 2]
            self.rowaddr(pair, D_C, mb * 32 + 8 * g, S_LDC, 4)
            t = self.tmp(4)
            self.read_bias(t, mb, g, nb)
            self.transpose4(t)
            self.store(4, VOC, t, pair, nb * 128)

    @staticmethod
    def e3_loads(d):
        """residual loads of drain step d into buffer set d % 2 (issued ONE STEP AHEAD: by body d-1, by `last` for d = 0), in the
        transposed layout of the stores: lane i of a quad = row i, 4 consecutive columns"""
        L = []
        for i, (mb, g, nb) in enumerate(step_units(d)):
            pair = ROWC if i % 2 == 0 else ROWD
            L += [f"s_mul_i32 {sr(ST + 4)}, {sr(S_LDR)}, {(mb * 32 + 8 * g) * 4}", f"s_add_u32 {sr(pair)}, {sr(D_R)}, {sr(ST + 4)}",
                  f"s_addc_u32 {sr(pair + 1)}, {sr(D_R + 1)}, 0"]
            L.append(("vm", f"global_load_dwordx4 {vr(RBUF + (d % 2) * RS + i * 4, 4)}, {vr(VOR)}, {sr(pair, 2)} offset:{nb * 128}", f"Rstep{d}"))
        return L

    def e3(self, units):
        d = self.d
        if d + 1 < NDRAIN:
            self.F += self.e3_loads(d + 1)
        self.F.append(("needprev", d))
        for i, (mb, g, nb) in enumerate(units):
            pair = (ROWA, ROWB)[i % 2]
            self.rowaddr(pair, D_C, mb * 32 + 8 * g, S_LDC, 4)
            t = self.tmp(4)
            self.read_bias(t, mb, g, nb)
            self.transpose4(t)
            for k in range(4):
                self.add(f"v_add_f32 {vr(t + k)}, {vr(t + k)}, {vr(RBUF + (d % 2) * RS + i * 4 + k)}")
            self.store(4, VOC, t, pair, nb * 128)

    def gelu4(self, x):
        """ds2_gelu (common.h) on v[x..x+3], the compiled instruction sequence of the other kernels, four chains interleaved"""
        A = self.add
        b = self.tmp(16)
        z, t, p, ee = ([b + 4 * j + c for c in range(4)] for j in range(4))
        msk = ("vcc", sr(ST + 2, 2), sr(ROWC, 2), sr(ROWD, 2))
        R4 = range(4)
        for c in R4: A(f"v_mul_f32_e64 {vr(z[c])}, |{vr(x + c)}|, {sr(G_C0)}")
        for c in R4: A(f"v_fma_f32 {vr(t[c])}, {vr(z[c])}, {sr(G_C1)}, 1.0")
        for c in R4: A(f"v_rcp_f32 {vr(t[c])}, {vr(t[c])}")
        for c in R4: A(f"v_mul_f32_e64 {vr(ee[c])}, {vr(z[c])}, -{vr(z[c])}")
        for c in R4: A(f"v_fmamk_f32 {vr(p[c])}, {vr(t[c])}, 0x3f87dc22, {vr(KC2)}")
        for c in R4: A(f"v_mul_f32 {vr(ee[c])}, 0x3fb8aa3b, {vr(ee[c])}")
        for c in R4: A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0x3fb5f0e3")
        for c in R4: A(f"v_exp_f32 {vr(ee[c])}, {vr(ee[c])}")
        for c in R4: A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0xbe91a98e")
        for c in R4: A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0x3e827906")
        for c in R4: A(f"v_mul_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}")
        for c in R4: A(f"v_mul_f32 {vr(p[c])}, 0.5, {vr(p[c])}")
        for c in R4: A(f"v_mul_f32 {vr(p[c])}, {vr(ee[c])}, {vr(p[c])}")
        for c in R4: A(f"v_cmp_le_f32_e64 {msk[c]}, 0, {vr(x + c)}")
        for c in R4: A(f"v_sub_f32 {vr(z[c])}, 1.0, {vr(p[c])}")
        for c in R4: A(f"v_cndmask_b32_e64 {vr(p[c])}, {vr(p[c])}, {vr(z[c])}, {msk[c]}")
        for c in R4: A(f"v_mul_f32 {vr(x + c)}, {vr(x + c)}, {vr(p[c])}")

    def e2(self, units):
        A = self.add
        for i, (mb, g, nb) in enumerate(units):
            ph, pl = (ROWA, ROWB)
            self.rowaddr(ph, D_H, mb * 32 + 8 * g, S_LDCP, 2)
            self.rowaddr(pl, D_L, mb * 32 + 8 * g, S_LDCP, 2)
            x = self.tmp(4)
            self.read_bias(x, mb, g, nb)
            if "nogelu" not in FLAGS:
                self.gelu4(x)
            w = self.tmp(16)
            hA, hB, lA, lB, f0, f1, f2, f3 = (w + j for j in range(8))
            n0, n1, o0, o1, xs, ys = (w + 8 + j for j in range(6))
            # bf16 hi / lo of rows (0, 1) -> A, rows (2, 3) -> B (the split of the other kernels: lo = bf16(v - float(hi)))
            A(f"v_cvt_pk_bf16_f32 {vr(hA)}, {vr(x)}, {vr(x + 1)}")
            A(f"v_cvt_pk_bf16_f32 {vr(hB)}, {vr(x + 2)}, {vr(x + 3)}")
            A(f"v_lshlrev_b32 {vr(f0)}, 16, {vr(hA)}")
            A(f"v_and_b32 {vr(f1)}, 0xffff0000, {vr(hA)}")
            A(f"v_lshlrev_b32 {vr(f2)}, 16, {vr(hB)}")
            A(f"v_and_b32 {vr(f3)}, 0xffff0000, {vr(hB)}")
            for k, f in enumerate((f0, f1, f2, f3)):
                A(f"v_sub_f32 {vr(f)}, {vr(x + k)}, {vr(f)}")
            A(f"v_cvt_pk_bf16_f32 {vr(lA)}, {vr(f0)}, {vr(f1)}")
            A(f"v_cvt_pk_bf16_f32 {vr(lB)}, {vr(f2)}, {vr(f3)}")
            for (pa, pb, pair) in ((hA, hB, ph), (lA, lB, pl)):
                o = self.tmp(2)          # (even-aligned pair: the 8-byte store's data)
                # stage 1 (lane ^ 1): even lanes keep row 0 / 2 with columns (c, c+1), odd lanes row 1 / 3 with columns (c-1, c)
                A(f"v_mov_b32_dpp {vr(n0)}, {vr(pa)} {XOR1}")
                A(f"v_mov_b32_dpp {vr(n1)}, {vr(pb)} {XOR1}")
                A(f"v_perm_b32 {vr(o0)}, {vr(n0)}, {vr(pa)}, {vr(SEL)}")
                A(f"v_perm_b32 {vr(o1)}, {vr(n1)}, {vr(pb)}, {vr(SEL)}")
                # stage 2 (lane ^ 2): lanes 0, 1 of a quad end with rows 0, 1 (A), lanes 2, 3 with rows 2, 3 (B), 4 columns each
                A(f"v_cndmask_b32_e64 {vr(xs)}, {vr(o1)}, {vr(o0)}, {sr(S_M2, 2)}")
                A("s_nop 1")
                A(f"v_mov_b32_dpp {vr(ys)}, {vr(xs)} {XOR2}")
                A(f"v_cndmask_b32_e64 {vr(o)}, {vr(o0)}, {vr(ys)}, {sr(S_M2, 2)}")
                A(f"v_cndmask_b32_e64 {vr(o + 1)}, {vr(ys)}, {vr(o1)}, {sr(S_M2, 2)}")
                self.store(2, VOP, o, pair, nb * 64)


# ---- the first form of the drain (kept for A/B runs, X4G_WIDE=0): no transposes - fp32 results leave as dword stores of 2 rows x
# 128 bytes, bf16 planes as dword stores after ONE lane-pair exchange (4 rows x 64 bytes per instruction)
def row_of(mb, r):
    return mb * 32 + (r & 3) + 8 * (r >> 2)


class DrainNarrow:
    """the filler stream of drain step d of the parked tile.  Items: instruction strings; ("vm", ins, name) VMEM ops; ("needvm", name)"""

    def __init__(self, d):
        self.F = []
        self.tix = 0
        self.d = d
        mb, sub = divmod(d, STEPS_PER_MB)
        pairs = [sub * RP + i for i in range(RP)]
        getattr(self, EPI)(mb, pairs)

    def tmp(self, n):
        if self.tix + n > NTMP:
            self.tix = 0
        b = T0 + self.tix
        self.tix += n
        return b

    def add(self, s):
        self.F.append(s)

    def rowaddr(self, pair, base, row, ld, esz):
        self.add(f"s_mul_i32 {sr(ST + 4)}, {sr(ld)}, {row * esz}")
        self.add(f"s_add_u32 {sr(pair)}, {sr(base)}, {sr(ST + 4)}")
        self.add(f"s_addc_u32 {sr(pair + 1)}, {sr(base + 1)}, 0")

    def store(self, voff, data, pair, off, name):
        if "nostore" in FLAGS:
            return
        self.F.append(("vm", f"global_store_dword {vr(voff)}, {vr(data)}, {sr(pair, 2)} offset:{off}{STMOD}", name))

    def e1(self, mb, pairs):
        k = 0
        for q in pairs:
            for r in (2 * q, 2 * q + 1):
                pair = (ROWA, ROWB, ROWC, ROWD)[k % 4]
                k += 1
                self.rowaddr(pair, D_C, row_of(mb, r), S_LDC, 4)
                for nb in range(NBW):
                    t = self.tmp(1)
                    self.add(f"v_accvgpr_read_b32 {vr(t)}, {park(mb, nb, r)}")
                    self.add(f"v_add_f32 {vr(t)}, {vr(t)}, {vr(BIAS + nb)}")
                    self.store(VOC, t, pair, nb * 128, "st")

    @staticmethod
    def e3_loads(d):
        mb, sub = divmod(d, STEPS_PER_MB)
        rows = [r for q in (sub * RP + i for i in range(RP)) for r in (2 * q, 2 * q + 1)]
        L = []
        for i, r in enumerate(rows):
            pair = ROWC if i % 2 == 0 else ROWD
            L += [f"s_mul_i32 {sr(ST + 4)}, {sr(S_LDR)}, {row_of(mb, r) * 4}", f"s_add_u32 {sr(pair)}, {sr(D_R)}, {sr(ST + 4)}",
                  f"s_addc_u32 {sr(pair + 1)}, {sr(D_R + 1)}, 0"]
            for nb in range(NBW):
                L.append(("vm", f"global_load_dword {vr(RBUF + (d % 2) * 8 + i * NBW + nb)}, {vr(VOR)}, {sr(pair, 2)} offset:{nb * 128}", f"Rstep{d}"))
        return L

    def e3(self, mb, pairs):
        d = self.d
        if d + 1 < NDRAIN:
            self.F += self.e3_loads(d + 1)
        self.F.append(("needprev", d))
        rows = [(q, r) for q in pairs for r in (2 * q, 2 * q + 1)]
        for i, (q, r) in enumerate(rows):
            pair = ROWA if i % 2 == 0 else ROWB
            self.rowaddr(pair, D_C, row_of(mb, r), S_LDC, 4)
            for nb in range(NBW):
                t = self.tmp(1)
                self.add(f"v_accvgpr_read_b32 {vr(t)}, {park(mb, nb, r)}")
                self.add(f"v_add_f32 {vr(t)}, {vr(t)}, {vr(BIAS + nb)}")
                self.add(f"v_add_f32 {vr(t)}, {vr(t)}, {vr(RBUF + (d % 2) * 8 + i * NBW + nb)}")
                self.store(VOC, t, pair, nb * 128, "st")

    def e2(self, mb, pairs):
        for i, q in enumerate(pairs):
            ph, pl = (ROWA, ROWB) if i % 2 == 0 else (ROWC, ROWD)
            self.rowaddr(ph, D_H, row_of(mb, 2 * q), S_LDCP, 2)
            self.rowaddr(pl, D_L, row_of(mb, 2 * q), S_LDCP, 2)
            for nb in range(NBW):
                b = self.tmp(20)
                x, z, t, p, ee, qq = ([b + 2 * j, b + 2 * j + 1] for j in range(6))      # [chain A, chain B]
                h, l, f0, f1, hn, ln, oh, ol = (b + 12 + j for j in range(8))
                A = self.add
                msk = ("vcc", sr(ST + 2, 2))
                for c, r in enumerate((2 * q, 2 * q + 1)):
                    A(f"v_accvgpr_read_b32 {vr(x[c])}, {park(mb, nb, r)}")
                for c in range(2):
                    A(f"v_add_f32 {vr(x[c])}, {vr(x[c])}, {vr(BIAS + nb)}")
                if "nogelu" not in FLAGS:
                    for c in range(2):
                        A(f"v_mul_f32_e64 {vr(z[c])}, |{vr(x[c])}|, {sr(G_C0)}")
                    for c in range(2):
                        A(f"v_fma_f32 {vr(t[c])}, {vr(z[c])}, {sr(G_C1)}, 1.0")
                    for c in range(2):
                        A(f"v_rcp_f32 {vr(t[c])}, {vr(t[c])}")
                    for c in range(2):
                        A(f"v_mul_f32_e64 {vr(ee[c])}, {vr(z[c])}, -{vr(z[c])}")
                    for c in range(2):
                        A(f"v_fmamk_f32 {vr(p[c])}, {vr(t[c])}, 0x3f87dc22, {vr(KC2)}")
                    for c in range(2):
                        A(f"v_mul_f32 {vr(ee[c])}, 0x3fb8aa3b, {vr(ee[c])}")
                    for c in range(2):
                        A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0x3fb5f0e3")
                    for c in range(2):
                        A(f"v_exp_f32 {vr(ee[c])}, {vr(ee[c])}")
                    for c in range(2):
                        A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0xbe91a98e")
                    for c in range(2):
                        A(f"v_fmaak_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}, 0x3e827906")
                    for c in range(2):
                        A(f"v_mul_f32 {vr(p[c])}, {vr(t[c])}, {vr(p[c])}")
                    for c in range(2):
                        A(f"v_mul_f32 {vr(p[c])}, 0.5, {vr(p[c])}")
                    for c in range(2):
                        A(f"v_mul_f32 {vr(p[c])}, {vr(ee[c])}, {vr(p[c])}")
                    for c in range(2):
                        A(f"v_cmp_le_f32_e64 {msk[c]}, 0, {vr(x[c])}")
                    for c in range(2):
                        A(f"v_sub_f32 {vr(qq[c])}, 1.0, {vr(p[c])}")
                    for c in range(2):
                        A(f"v_cndmask_b32_e64 {vr(p[c])}, {vr(p[c])}, {vr(qq[c])}, {msk[c]}")
                    for c in range(2):
                        A(f"v_mul_f32 {vr(x[c])}, {vr(x[c])}, {vr(p[c])}")
                if MX:
                    # MX activation planes of the two rows (common.h): plane 1 = fp16 pair (MODE.FP16_OVFL is set: the conversions
                    # SATURATE), plane 2 = (e4m3(v 2^-EA) | e4m3((v - fp16(v)) 2^LA) << 8) per element; columns paired between lanes below
                    A(f"v_cvt_pk_f16_f32 {vr(h)}, {vr(x[0])}, {vr(x[1])}")
                    A(f"v_cvt_scalef32_pk_fp8_f32 {vr(hn)}, {vr(x[0])}, {vr(x[1])}, {sr(S_SCV)}")        # value bytes (v0, v1) in the low half
                    A(f"v_cvt_f32_f16_e32 {vr(f0)}, {vr(h)}")
                    A(f"v_cvt_f32_f16_sdwa {vr(f1)}, {vr(h)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
                    A(f"v_sub_f32 {vr(f0)}, {vr(x[0])}, {vr(f0)}")
                    A(f"v_sub_f32 {vr(f1)}, {vr(x[1])}, {vr(f1)}")
                    A(f"v_cvt_scalef32_pk_fp8_f32 {vr(ln)}, {vr(f0)}, {vr(f1)}, {sr(S_SCR)}")            # remainder bytes (r0, r1)
                    A(f"v_perm_b32 {vr(l)}, {vr(ln)}, {vr(hn)}, {sr(S_PSEL)}")                           # (v0, r0, v1, r1)
                else:
                    # bf16 hi / lo of the two rows, columns paired between neighbouring lanes (module docstring)
                    A(f"v_cvt_pk_bf16_f32 {vr(h)}, {vr(x[0])}, {vr(x[1])}")
                    A(f"v_lshlrev_b32 {vr(f0)}, 16, {vr(h)}")
                    A(f"v_and_b32 {vr(f1)}, 0xffff0000, {vr(h)}")
                    A(f"v_sub_f32 {vr(f0)}, {vr(x[0])}, {vr(f0)}")
                    A(f"v_sub_f32 {vr(f1)}, {vr(x[1])}, {vr(f1)}")
                    A(f"v_cvt_pk_bf16_f32 {vr(l)}, {vr(f0)}, {vr(f1)}")
                A(f"v_mov_b32_dpp {vr(hn)}, {vr(h)} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                A(f"v_perm_b32 {vr(oh)}, {vr(hn)}, {vr(h)}, {vr(SEL)}")
                self.store(VOP, oh, ph, nb * 64, "st")
                A(f"v_mov_b32_dpp {vr(ln)}, {vr(l)} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                A(f"v_perm_b32 {vr(ol)}, {vr(ln)}, {vr(l)}, {vr(SEL)}")
                self.store(VOP, ol, pl, nb * 64, "st")



WIDE = os.environ.get("X4G_WIDE", "1" if (KSUB == 2 and EPI != "e2") else _DEF_WIDE) != "0"


def drain_items(d):
    if "nodrain" in FLAGS:
        return []
    if WIDE:
        return Drain(d).F
    if KSUB == 2:                        # a 64-deep body carries two of the 16 narrow steps (e2 only: no state between steps)
        assert EPI == "e2"
        return DrainNarrow(2 * d).F + DrainNarrow(2 * d + 1).F
    return DrainNarrow(d).F


def n_loads(d):
    """residual loads of e3 step d"""
    return len(step_units(d)) if WIDE else 2 * RP * NBW


def n_stores(d):
    return len(step_units(d)) if WIDE else 2 * RP * NBW


# ------------------------------------------------------------------------------------------------ one K-tile body
body_n = [0]


def body(kind, d=None, hist=((0, 0), (0, 0)), prev_after=0):
    """kind: 'plain' | 'drain' (step d) | 'last'.  hist[j] = (VMEM ops, VMEM ops after the last DMA piece) of the body j + 1 before
    this one (minimum over its possible predecessors); prev_after: VMEM ops the previous body issued after the residual loads it
    issued for this step (e3).  Returns this body's (VMEM ops, VMEM ops after its last DMA piece)."""
    n = body_n[0]
    body_n[0] += 1
    lg = Stream(POST_BAR)
    vm_issued = [0]          # VMEM ops of this body, program order
    vm_names = []
    dma_last = [None]

    def emit_vm(ins, name):
        e(ins)
        vm_names.append(name)
        vm_issued[0] += 1

    def need_lg(name):
        c = lg.need(name)
        if c is not None:
            e(f"s_waitcnt lgkmcnt({min(c, 15)})")

    # --- filler queues.  DMA block: [cursor advance] [destinations] [pieces] [stage rotation]; with DMA_POST it is queued at the
    #     barrier slot (the barrier frees the stage it overwrites), else at the top of the body.
    Qdma = [("blk", advance_block(n))] + dma_dst_setup()
    if "nodma" not in FLAGS:
        for m0w, cp in dma_pieces():
            Qdma.append(m0w)
            Qdma.append(("dma", cp))
    Qdma += rotate_dma_dst()
    Q = []
    if kind == "last" and DMA_POST:
        Q += last_setup()                    # (older than this body's DMA pieces: the next body waits for them with the pieces in flight)
    if not DMA_POST:
        Q += Qdma
    Q += rotate_read_delta()
    if kind == "drain":
        if d == 0:
            Q.append(("waitvm_setup",))      # the bias (+ residual) loads of `last`
        items = drain_items(d)
        if STDEFER and DMA_POST and WIDE:
            # the result stores of this step are issued BEHIND this body's DMA pieces: the next barrier then waits for K tile t+2 with
            # them still in flight (in-order VMEM completion: a store issued before the pieces would have to retire within one body)
            Qdma += [it for it in items if not isinstance(it, str) and it[0] == "vm" and it[2] == "st"]
            items = [it for it in items if isinstance(it, str) or not (it[0] == "vm" and it[2] == "st")]
        Q += items
    pos = [0]
    debt = [0.0]

    def emit_one():
        it = Q[pos[0]]
        pos[0] += 1
        if isinstance(it, str):
            e(it)
            return cost(it)
        if it[0] == "blk":
            for s in it[1]:
                e(s)
            return 6 * ISSUE
        if it[0] == "dma":
            e("s_nop 0")
            emit_vm(it[1], "dma")
            dma_last[0] = vm_issued[0]
            return 2 * ISSUE
        if it[0] == "vm":
            emit_vm(it[1], it[2])
            return ISSUE
        if it[0] == "needvm":
            idx = max(i for i, nm in enumerate(vm_names) if nm == it[1])
            e(f"s_waitcnt vmcnt({min(vm_issued[0] - idx - 1, 63)})")
            return ISSUE
        if it[0] == "needprev":              # the loads the previous body issued for this step (it issued prev_after VMEM ops after them)
            e(f"s_waitcnt vmcnt({min(prev_after + vm_issued[0], 63)})")
            return ISSUE
        if it[0] == "waitvm_setup":          # `last` issued its set-up loads before (DMA_POST) / after its NP DMA pieces
            e(f"s_waitcnt vmcnt({min((NP if DMA_POST else 0) + vm_issued[0], 63)})")
            return ISSUE
        raise ValueError(it)

    def fill(budget):
        debt[0] += budget
        while pos[0] < len(Q) and debt[0] > 0:
            debt[0] -= emit_one()

    def flush():
        while pos[0] < len(Q):
            emit_one()

    # --- MFMA slots.  bf16x3: NQ sub-steps of 3 NB MFMAs; sub-step q multiplies fragment set X (q even) / Y (q odd).
    #     MX: NQ sub-steps of NB fp16 MFMAs (a_hi w_hi), then per block TWO scaled fp8 MFMAs, each the two cross terms of 32 k's
    def sub_slots(q, block_major):
        setbase = FX if q % 2 == 0 else FY
        blocks = [(mb, nb) for mb in range(MBW) for nb in range(NBW)]
        if MX:
            return [(mb, nb, "ah", "wh", setbase, q) for mb, nb in blocks]
        terms = [("al", "wh"), ("ah", "wl"), ("ah", "wh")]
        if block_major:
            return [(mb, nb, ka, kb, setbase, q) for mb, nb in blocks for ka, kb in terms]
        return [(mb, nb, ka, kb, setbase, q) for ka, kb in terms for mb, nb in blocks]

    slots = []
    for q in range(NQ):
        slots += sub_slots(q, kind == "last" and q == NQ - 1)
    if MX:
        for j in range(2):
            slots += [(mb, nb, "a8", "w8", None, NQ + j) for mb in range(MBW) for nb in range(NBW)]
    assert len(slots) == NSLOT
    rq = {q: list(y_order()) for q in range(1, NQ)}      # fragment reads of sub-step q, issued behind the MFMAs of sub-step q - 1
    r8 = list(f8_order()) if MX else []                  # MX: plane-2 reads of THIS K tile, issued behind the fp16 MFMAs before the barrier
    xq = None
    park_q = []              # (slot index after which block b may be parked)
    first = kind == "drain" and d == 0
    for si, (mb, nb, ka, kb, sb, q) in enumerate(slots):
        pre = "X" if sb == FX else "Y"
        if si == BAR_SLOT:
            # K tile t+1 landed (this wave's pieces; younger VMEM ops stay in flight), everybody past its reads of this stage's
            # predecessor.  VMEM ops younger than the last piece of K tile t+1 (issued LOOK - 1 bodies ago):
            if not DMA_POST:
                assert not [x for x in Q[pos[0]:] if not isinstance(x, str) and x[0] == "dma"], "DMA pieces must precede the barrier slot"
            assert not r8 and not any(rq.values()), "every read of this stage must be issued before its barrier"
            nvm = hist[LOOK - 2][1] + sum(hist[j][0] for j in range(LOOK - 2)) + vm_issued[0]
            e(f"s_waitcnt vmcnt({min(nvm, 63)})")
            e("s_waitcnt lgkmcnt(0)")           # (this wave's reads of the stage the next DMA overwrites)
            lg.wait_all()
            if "nobarrier" not in FLAGS:
                e("s_barrier")
            e(f"v_add_u32 {vr(RA)}, {sr(S_RDELTA)}, {vr(RA)}")
            e(f"v_add_u32 {vr(RW)}, {sr(S_RDELTA)}, {vr(RW)}")
            xq = [("X", k, i) for k, i in x_order()]
            if MX and DMA_PACE:
                debt[0] = min(debt[0], 0.0)     # the DMA block is PACED under the fp16 / long fp8 MFMAs behind the barrier, not dumped in one gap
            elif MX:
                debt[0] = max(debt[0], 10000.0) # ... or issued in ONE burst right behind the barrier (the window of a piece ends at the next barrier)
            if DMA_POST:
                Q[pos[0]:pos[0]] = Qdma
        b = mb * NBW + nb
        is8 = ka == "a8"
        if is8:
            j = q - NQ
            for nm in (f"F8a{mb}{2 * j}", f"F8a{mb}{2 * j + 1}", f"F8w{nb}{2 * j}", f"F8w{nb}{2 * j + 1}"):
                need_lg(nm)
            if "nomfma" not in FLAGS:
                e(f"v_mfma_scale_f32_32x32x64_f8f6f4 {acc(b)}, {f8('a', mb, j)}, {f8('w', nb, j)}, {acc(b)}, {vr(SCA)}, {vr(SCB)} op_sel_hi:[0,0,0]")
        else:
            need_lg(f"{pre}{ka}{mb}")
            need_lg(f"{pre}{kb}{nb}")
            c = "0" if (first and q == 0 and (MX or (ka, kb) == ("al", "wh"))) else acc(b)
            if "nomfma" not in FLAGS:
                e(f"v_mfma_f32_32x32x16_{'f16' if MX else 'bf16'} {acc(b)}, {frag(sb, ka, mb)}, {frag(sb, kb, nb)}, {c}")
        spent = 0
        if q + 1 < NQ and rq[q + 1]:
            # (MX: a sub-step has NB slots for MBW + NBW reads of the next sub-step and its share of the plane-2 reads)
            early = RD_EARLY if (MX and q + 1 == NQ - 1) else 0      # (the sub-step in front of the barrier)
            for _ in range(2 if (MX and len(rq[q + 1]) > NB - 1 - early - si % NB) else 1):
                if rq[q + 1]:
                    k, i = rq[q + 1].pop(0)
                    nset, npre = (FY, "Y") if (q + 1) % 2 else (FX, "X")
                    if "noread" not in FLAGS:
                        e(read_ins(nset, k, i, q + 1))
                    lg.issue(f"{npre}{k}{i}")
                    spent += ISSUE
        if r8 and si < BAR_SLOT:
            left = max(BAR_SLOT - RD_EARLY - si, 1)    # slots up to the barrier (less RD_EARLY), this one included
            for _ in range((len(r8) + left - 1) // left):
                k, i, qp = r8.pop(0)
                if "noread" not in FLAGS:
                    e(read8_ins(k, i, qp))
                lg.issue(f"F8{k}{i}{qp}")
                spent += ISSUE
        if xq:
            for _ in range(2 if len(xq) > (NSLOT - 1 - si) else 1):
                if xq:
                    _, k, i = xq.pop(0)
                    if "noread" not in FLAGS:
                        e(read_ins(FX, k, i, 0))
                    lg.issue(f"X{k}{i}")
                    spent += ISSUE
        last_of_block = (is8 and q == NQ + 1) if MX else (q == NQ - 1 and (ka, kb) == ("ah", "wh"))
        if kind == "last" and last_of_block:
            park_q.append((si + 3, b))          # three more MFMAs (>= 96 cycles) before the block's accumulators are read
        while park_q and park_q[0][0] <= si:
            _, pb = park_q.pop(0)
            for r in range(16):
                e(f"v_accvgpr_mov_b32 {ar(128 + 16 * pb + r)}, {ar(16 * pb + r)}")
            spent += 16 * ISSUE
        fill((GAP8 if is8 else GAP) - spent)
    assert not xq and not any(rq.values()), (xq, rq)
    flush()
    for q in range(1, NQ):
        e(f"v_add_u32 {vr(RA + q)}, {sr(S_RDELTA)}, {vr(RA + q)}")
        e(f"v_add_u32 {vr(RW + q)}, {sr(S_RDELTA)}, {vr(RW + q)}")
    if kind == "last":
        e("s_nop 15")
        while park_q:
            _, pb = park_q.pop(0)
            for r in range(16):
                e(f"v_accvgpr_mov_b32 {ar(128 + 16 * pb + r)}, {ar(16 * pb + r)}")
        if not DMA_POST:
            Q[:] = last_setup()
            pos[0] = 0
            flush()
    return (vm_issued[0], vm_issued[0] - (dma_last[0] or 0))

def body_mx2(kind, d=None, prev_after=0):
    """The MX K-tile body with the stage's two operand planes re-filled SEPARATELY (round 6, second form).  With one barrier per K tile
    and two stages only ONE tile's LDS-DMA is ever in flight, issued behind barrier t and needed at barrier t+1: the L2 -> LDS stream
    idles from its completion to the next barrier and every tile pays the L2 latency (measured: 0.72 us of the 1.22 us a tile's 80 KB
    take at the CU's share of the L2 rate stay exposed, profiles/r06_mx_time.txt).  Here:
        q0 q1 | B1 | F F | q2 | B2 | q3        (q: 6 fp16 MFMAs of a 16-deep sub-step, F: 6 scaled fp8 MFMAs)
      * the plane-2 (fp8) fragments of tile t are read behind q0 / q1; B1 = every wave is past them -> the plane-2 region of the stage
        is free and plane 2 of tile t+2 is issued under the long fp8 MFMAs (window: to B2 of the NEXT body, 1.6 K tiles);
      * the fp16 fragments of q2 / q3 are read behind q1 / F; B2 = plane 1 of the stage is free and tile t+1 has landed -> plane 1 of
        tile t+2 and the first fragments of tile t+1 are issued under q3.
    The vmcnt of B2 is counted inside the body: everything older than this body's own plane-2 pieces must be complete.
    Returns (VMEM ops of the body, VMEM ops issued after its last bias / residual load or None)."""
    n = body_n[0]
    body_n[0] += 1
    lg = Stream(POST_BAR)
    vm_issued, vm_names = [0], []
    dma2_start = [None]

    def emit_vm(ins, name):
        e(ins)
        vm_names.append(name)
        vm_issued[0] += 1

    def need_lg(name):
        c = lg.need(name)
        if c is not None:
            e(f"s_waitcnt lgkmcnt({min(c, 15)})")

    nodma = "nodma" in FLAGS
    Qd2 = [("blk", advance_block(n))] + dma_dst_setup()          # behind B1: cursor -> tile t+2, destinations, plane 2
    Qd1 = []                                                     # behind B2: plane 1, stage rotation
    if not nodma:
        for m0w, cp in dma_pieces((1,)):
            Qd2 += [m0w, ("dma2", cp)]
        for m0w, cp in dma_pieces((0,)):
            Qd1 += [m0w, ("dma", cp)]
    Qd1 += rotate_dma_dst()
    Q = []
    if kind == "last":
        Q += last_setup()
    Q += rotate_read_delta()
    if kind == "drain":
        if d == 0:
            Q.append(("waitvm_setup",))
        Q += drain_items(d)
    pos, debt = [0], [0.0]

    def emit_one():
        it = Q[pos[0]]
        pos[0] += 1
        if isinstance(it, str):
            e(it)
            return cost(it)
        if it[0] == "blk":
            for s_ in it[1]:
                e(s_)
            return 6 * ISSUE
        if it[0] in ("dma", "dma2"):
            e("s_nop 0")
            if it[0] == "dma2" and dma2_start[0] is None:
                dma2_start[0] = vm_issued[0]
            emit_vm(it[1], it[0])
            return 2 * ISSUE
        if it[0] == "vm":
            emit_vm(it[1], it[2])
            return ISSUE
        if it[0] == "needvm":
            idx = max(i for i, nm in enumerate(vm_names) if nm == it[1])
            e(f"s_waitcnt vmcnt({min(vm_issued[0] - idx - 1, 63)})")
            return ISSUE
        if it[0] in ("needprev", "waitvm_setup"):      # loads the PREVIOUS body issued (it issued prev_after VMEM ops behind them)
            e(f"s_waitcnt vmcnt({min(prev_after + vm_issued[0], 63)})")
            return ISSUE
        raise ValueError(it)

    def fill(budget):
        debt[0] += budget
        while pos[0] < len(Q) and debt[0] > 0:
            debt[0] -= emit_one()

    # ---- slots: (kind, mb, nb, q | j)
    blocks = [(mb, nb) for mb in range(MBW) for nb in range(NBW)]
    S = [("h", mb, nb, 0) for mb, nb in blocks] + [("h", mb, nb, 1) for mb, nb in blocks]
    S += [("f", mb, nb, j) for j in range(2) for mb, nb in blocks]
    S += [("h", mb, nb, 2) for mb, nb in blocks] + [("h", mb, nb, 3) for mb, nb in blocks]
    assert len(S) == NSLOT and NQ == 4
    B1, B2 = 2 * NB, 5 * NB
    pre = "XYZW"
    setof = lambda q: FX if q % 2 == 0 else FY                   # noqa: E731
    # read queues: (slot from which it may be issued, instruction, stream name)
    R = []
    r8 = f8_order()
    for i_, (k, i, qp) in enumerate(r8):                         # plane 2 of THIS tile: slots 0.., three per slot
        R.append((i_ // 3, read8_ins(k, i, qp), f"F8{k}{i}{qp}"))
    for i_, (k, i) in enumerate(y_order()):                      # q1's fragments (set Y): behind the first MFMAs of q0
        R.append((0, read_ins(FY, k, i, 1), f"Y{k}{i}"))
    for i_, (k, i) in enumerate(y_order()):                      # q2's fragments (set X again): X is free once q0 has issued
        R.append((NB, read_ins(FX, k, i, 2), f"Z{k}{i}"))
    for i_, (k, i) in enumerate(y_order()):                      # q3's fragments (set Y again): behind B1, under the fp8 MFMAs
        R.append((B1, read_ins(FY, k, i, 3), f"W{k}{i}"))
    for i_, (k, i) in enumerate(x_order()):                      # the NEXT tile's q0 fragments: behind B2
        R.append((B2, read_ins(FX, k, i, 0), f"X{k}{i}"))
    R.sort(key=lambda r: r[0])
    rp = [0]
    park_q = []
    first = kind == "drain" and d == 0

    def barrier(wait_vm, lg_last=None):
        if wait_vm is not None:
            e(f"s_waitcnt vmcnt({min(wait_vm, 63)})")
        if lg_last is None:
            e("s_waitcnt lgkmcnt(0)")
            lg.wait_all()
        else:                      # (B1: this wave's plane-2 reads only - the younger fp16 reads of q2 stay in flight)
            need_lg(lg_last)
        if "nobarrier" not in FLAGS:
            e("s_barrier")

    for si, (kd, mb, nb, q) in enumerate(S):
        if si == B1:
            assert all(r[0] >= B1 for r in R[rp[0]:]), "plane-2 reads must be issued before B1"
            k_, i_, qp_ = r8[-1]
            barrier(None, f"F8{k_}{i_}{qp_}")
            debt[0] = min(debt[0], 0.0)
            Q[pos[0]:pos[0]] = Qd2
        if si == B2:
            while any((not isinstance(x, str)) and x[0] == "dma2" for x in Q[pos[0]:]):     # (never taken at the default gaps)
                emit_one()
            assert all(r[0] >= B2 for r in R[rp[0]:]), "every read of this stage must be issued before B2"
            # tile t+1 (both planes) landed: everything older than this body's plane-2 pieces is complete
            barrier(None if (nodma or dma2_start[0] is None) else vm_issued[0] - dma2_start[0])
            e(f"v_add_u32 {vr(RA)}, {sr(S_RDELTA)}, {vr(RA)}")
            e(f"v_add_u32 {vr(RW)}, {sr(S_RDELTA)}, {vr(RW)}")
            debt[0] = min(debt[0], 0.0)
            Q[pos[0]:pos[0]] = Qd1
        b = mb * NBW + nb
        if kd == "f":
            for nm in (f"F8a{mb}{2 * q}", f"F8a{mb}{2 * q + 1}", f"F8w{nb}{2 * q}", f"F8w{nb}{2 * q + 1}"):
                need_lg(nm)
            if "nomfma" not in FLAGS:
                e(f"v_mfma_scale_f32_32x32x64_f8f6f4 {acc(b)}, {f8('a', mb, q)}, {f8('w', nb, q)}, {acc(b)}, {vr(SCA)}, {vr(SCB)} op_sel_hi:[0,0,0]")
        else:
            need_lg(f"{pre[q]}ah{mb}")
            need_lg(f"{pre[q]}wh{nb}")
            c = "0" if (first and q == 0) else acc(b)
            if "nomfma" not in FLAGS:
                e(f"v_mfma_f32_32x32x16_f16 {acc(b)}, {frag(setof(q), 'ah', mb)}, {frag(setof(q), 'wh', nb)}, {c}")
        spent = 0
        for _ in range(3 if si < B1 else 2):
            if rp[0] < len(R) and R[rp[0]][0] <= si:
                _, ins, nm = R[rp[0]]
                rp[0] += 1
                if "noread" not in FLAGS:
                    e(ins)
                lg.issue(nm)
                spent += ISSUE
        if kind == "last" and kd == "h" and q == NQ - 1:
            park_q.append((si + 3, b))
        while park_q and park_q[0][0] <= si:
            _, pb = park_q.pop(0)
            for r in range(16):
                e(f"v_accvgpr_mov_b32 {ar(128 + 16 * pb + r)}, {ar(16 * pb + r)}")
            spent += 16 * ISSUE
        fill((GAP8 if kd == "f" else GAP) - spent)
    assert rp[0] == len(R), (rp[0], len(R))
    while pos[0] < len(Q):
        emit_one()
    for q in range(1, NQ):
        e(f"v_add_u32 {vr(RA + q)}, {sr(S_RDELTA)}, {vr(RA + q)}")
        e(f"v_add_u32 {vr(RW + q)}, {sr(S_RDELTA)}, {vr(RW + q)}")
    if kind == "last":
        e("s_nop 15")
        while park_q:
            _, pb = park_q.pop(0)
            for r in range(16):
                e(f"v_accvgpr_mov_b32 {ar(128 + 16 * pb + r)}, {ar(16 * pb + r)}")
    loads = [i for i, nm in enumerate(vm_names) if nm == "bias" or nm.startswith("Rstep")]
    return vm_issued[0], (vm_issued[0] - max(loads) - 1) if loads else None


def last_setup():
    """drain bases and bias (+ the residual loads of step 0) of the tile this body parks (CUR), CUR <- NXT: item list"""
    L = []
    for base, arg, off in ((D_C, S_C, CUR), (D_R, S_R, CUR + 1), (D_H, S_CH, CUR + 2), (D_L, S_CL, CUR + 2)):
        if MX and EPI == "e2" and base in (D_C, D_R):
            continue                           # (their registers hold the MX conversion constants)
        wv = {CUR: W_C, CUR + 1: W_R, CUR + 2: W_P}[off]
        if "stsame" in FLAGS:                  # (timing experiment: every tile's result goes to the first tile's place)
            L.append(f"s_mov_b32 {sr(ST)}, {sr(wv)}")
        else:
            L.append(f"s_add_u32 {sr(ST)}, {sr(off)}, {sr(wv)}")
        L += [f"s_add_u32 {sr(base)}, {sr(arg)}, {sr(ST)}", f"s_addc_u32 {sr(base + 1)}, {sr(arg + 1)}, 0"]
    L += [f"s_add_u32 {sr(ST)}, {sr(CUR + 3)}, {sr(W_B)}", f"s_add_u32 {sr(ROWA)}, {sr(S_BIAS)}, {sr(ST)}",
          f"s_addc_u32 {sr(ROWA + 1)}, {sr(S_BIAS + 1)}, 0", f"v_lshlrev_b32 {vr(T0)}, 2, {vr(L31)}"]
    for nb in range(NBW):
        L.append(("vm", f"global_load_dword {vr(BIAS + nb)}, {vr(T0)}, {sr(ROWA, 2)} offset:{nb * 128}", "bias"))
    if EPI == "e3" and "nodrain" not in FLAGS:
        L += (Drain if WIDE else DrainNarrow).e3_loads(0)
    for i in range(4):
        L.append(f"s_mov_b32 {sr(CUR + i)}, {sr(NXT + i)}")
    return L


# ------------------------------------------------------------------------------------------------ prologue / tail
def prologue():
    t0, t1, t2, t3 = T0, T0 + 1, T0 + 2, T0 + 3
    e("s_mov_b32 s90, %[klo]")
    e("s_mov_b32 s91, %[khi]")
    e(f"s_load_dwordx16 {sr(36, 16)}, s[90:91], 0x0")
    e(f"s_load_dwordx8 {sr(52, 8)}, s[90:91], 0x40")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_mov_b32 {sr(G_C0)}, 0x3f3504f3")
    e(f"s_mov_b32 {sr(G_C1)}, 0x3ea7ba05")
    e(f"v_mov_b32 {vr(KC2)}, 0xbfba00e3")
    if MX:      # E8M0 scale bytes (byte 0, op_sel 0): the A operand carries the common factor 2^(EA - LW) of both cross terms, W is 1.0
        e(f"v_mov_b32 {vr(SCA)}, {MX_SCALE_A}")
        e(f"v_mov_b32 {vr(SCB)}, 127")
    e(f"v_and_b32 {vr(L31)}, 31, %[lane]")
    e(f"v_lshrrev_b32 {vr(HALF)}, 5, %[lane]")
    e("s_lshr_b32 s92, %[wave], 1")          # wm
    e("s_and_b32 s93, %[wave], 1")           # wn
    # LDS read bases: row (wave row 32 MBW wm + l31) x RBYTES + (chunk ^ swizzle) x 16, chunk = 2 q + half of sub-step q; the swizzle is
    # (row >> 2) & 3 on 64-byte rows, (row >> 1) & 7 on 128-byte rows (conflict-free over the 16-lane groups of ds_read_b128)
    e(f"v_lshrrev_b32 {vr(t0)}, {2 if KSUB == 1 else 1}, {vr(L31)}")
    e(f"v_and_b32 {vr(t0)}, {3 if KSUB == 1 else 7}, {vr(t0)}")            # sw
    e(f"s_mul_i32 s94, s92, {32 * MBW * RBYTES}")
    e("s_add_u32 s94, s94, %[ldsb]")
    e(f"s_mul_i32 s95, s93, {32 * NBW * RBYTES}")
    e("s_add_u32 s95, s95, %[ldsb]")
    e(f"s_add_u32 s95, s95, {2 * PA_B}")
    e(f"v_lshlrev_b32 {vr(t1)}, {6 if KSUB == 1 else 7}, {vr(L31)}")
    for q in range(NQ):
        e(f"v_or_b32 {vr(t2)}, {2 * q}, {vr(HALF)}")
        e(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(t0)}")
        e(f"v_lshl_add_u32 {vr(t2)}, {vr(t2)}, 4, {vr(t1)}")
        e(f"v_add_u32 {vr(RA + q)}, s94, {vr(t2)}")
        e(f"v_add_u32 {vr(RW + q)}, s95, {vr(t2)}")
    # LDS-DMA lane offsets.  64-byte rows: a piece is 16 rows x 64 B, lane -> (row lane >> 2, chunk (lane & 3) ^ ((lane >> 4) & 3)).
    # 128-byte rows: 8 rows x 128 B = full L2 lines, lane -> (row lane >> 3, chunk (lane & 7) ^ ((row >> 1) & 7)); the pieces of a wave
    # start at multiples of 8 rows and their count per wave is even, so (row >> 1) & 7 = (4 (j & 1) + (lane >> 4)) & 7 for piece j
    if KSUB == 1:
        e(f"v_lshrrev_b32 {vr(t0)}, 4, %[lane]")
        e(f"v_and_b32 {vr(t0)}, 3, {vr(t0)}")
        e(f"v_and_b32 {vr(t1)}, 3, %[lane]")
        e(f"v_xor_b32 {vr(t0)}, {vr(t0)}, {vr(t1)}")
        e(f"v_lshlrev_b32 {vr(t0)}, 4, {vr(t0)}")                          # chunk * 16
        e(f"v_lshrrev_b32 {vr(t1)}, 2, %[lane]")                           # row in the piece
        if "p128" in FLAGS:              # (timing experiment: pieces of 8 rows x 128 B - what a 64-deep K tile would fetch)
            e(f"v_and_b32 {vr(t0)}, 7, %[lane]")
            e(f"v_lshlrev_b32 {vr(t0)}, 4, {vr(t0)}")
            e(f"v_lshrrev_b32 {vr(t1)}, 3, %[lane]")
    else:
        assert PA_N % 2 == 0 and PW_N % 2 == 0
        e(f"v_lshrrev_b32 {vr(t0)}, 4, %[lane]")
        e(f"v_and_b32 {vr(t1)}, 7, %[lane]")
        e(f"v_xor_b32 {vr(t0)}, {vr(t0)}, {vr(t1)}")
        e(f"v_lshlrev_b32 {vr(t0)}, 4, {vr(t0)}")                          # chunk * 16 of the even pieces (odd pieces: ^ 64)
        e(f"v_lshrrev_b32 {vr(t1)}, 3, %[lane]")
    rows_pp = 8 if (KSUB == 2 or "p128" in FLAGS) else 16
    for cnt, ld, dst in ((PA_N, S_LDA, DOA), (PW_N, S_LDW, DOW)):
        e(f"s_mul_i32 s94, %[wave], {cnt * (16 // KSUB)}")
        e(f"v_add_u32 {vr(t2)}, s94, {vr(t1)}")
        e(f"s_lshl_b32 s94, {sr(ld)}, 1")
        e(f"v_mul_lo_u32 {vr(t2)}, {vr(t2)}, s94")
        e(f"v_add_u32 {vr(dst)}, {vr(t2)}, {vr(t0)}")
        e(f"s_mul_i32 s94, {sr(ld)}, {2 * rows_pp}")                       # one piece further down
        for j in range(1, cnt):
            e(f"v_add_u32 {vr(dst + j)}, s94, {vr(dst + j - 1)}")
            if KSUB == 2:
                e(f"v_xor_b32 {vr(dst + j)}, 64, {vr(dst + j)}")
    if WIDE:
        # store lane offsets (layout AFTER the quad transposes): lane -> row 4 half + (l31 & 3), columns (l31 & 28) .. + 3
        e(f"v_and_b32 {vr(t0)}, 3, {vr(L31)}")
        e(f"v_lshl_add_u32 {vr(t1)}, {vr(HALF)}, 2, {vr(t0)}")                 # 4 half + (l31 & 3)
        e(f"v_and_b32 {vr(t2)}, 28, {vr(L31)}")
        for vo, ld, sh in ((VOC, S_LDC, 2), (VOR, S_LDR, 2), (VOP, S_LDCP, 1)):
            e(f"s_lshl_b32 s94, {sr(ld)}, {sh}")                               # bytes per row
            e(f"v_mul_lo_u32 {vr(t3)}, {vr(t1)}, s94")
            e(f"v_lshl_add_u32 {vr(vo)}, {vr(t2)}, {sh}, {vr(t3)}")
    if not WIDE:        # lane = column: row 4 half, column l31 (fp32); row 4 half + (l31 & 1), columns l31 & 30 (planes)
        for vo, ld in ((VOC, S_LDC), (VOR, S_LDR)):
            e(f"s_lshl_b32 s94, {sr(ld)}, 4")
            e(f"v_mul_lo_u32 {vr(t2)}, {vr(HALF)}, s94")
            e(f"v_lshl_add_u32 {vr(vo)}, {vr(L31)}, 2, {vr(t2)}")
        e(f"v_and_b32 {vr(t0)}, 1, {vr(L31)}")
        e(f"v_lshl_add_u32 {vr(t1)}, {vr(HALF)}, 2, {vr(t0)}")
        e(f"s_lshl_b32 s94, {sr(S_LDCP)}, 1")
        e(f"v_mul_lo_u32 {vr(t1)}, {vr(t1)}, s94")
        e(f"v_and_b32 {vr(t2)}, 30, {vr(L31)}")
        e(f"v_lshl_add_u32 {vr(VOP)}, {vr(t2)}, 1, {vr(t1)}")
    # selector of the first exchange stage of the bf16 planes (lane ^ 1), lane masks of the transposes
    e(f"v_and_b32 {vr(t0)}, 1, {vr(L31)}")
    e(f"v_mov_b32 {vr(t1)}, 0x05040100")
    e(f"v_mov_b32 {vr(t2)}, 0x03020706")
    e(f"v_cmp_eq_u32 vcc, 1, {vr(t0)}")
    e("s_nop 1")
    e(f"v_cndmask_b32 {vr(SEL)}, {vr(t1)}, {vr(t2)}, vcc")
    e(f"s_mov_b32 {sr(S_M1)}, 0xaaaaaaaa")
    e(f"s_mov_b32 {sr(S_M1 + 1)}, 0xaaaaaaaa")
    e(f"s_mov_b32 {sr(S_M2)}, 0xcccccccc")
    e(f"s_mov_b32 {sr(S_M2 + 1)}, 0xcccccccc")
    # this wave's offsets inside a tile
    for dst, ld, esz in ((W_C, S_LDC, 4), (W_R, S_LDR, 4), (W_P, S_LDCP, 2)):
        e(f"s_mul_i32 s94, s92, {32 * MBW}")
        e(f"s_mul_i32 s94, s94, {sr(ld)}")
        e(f"s_mul_i32 s95, s93, {32 * NBW}")
        e("s_add_u32 s94, s94, s95")
        e(f"s_mul_i32 {sr(dst)}, s94, {esz}")
    e(f"s_mul_i32 {sr(W_B)}, s93, {32 * NBW * 4}")
    # loop state
    e(f"s_sub_u32 {sr(S_NKM1)}, {sr(S_NK)}, 1")
    e(f"s_add_u32 {sr(S_LEND)}, %[ldsb], {NSTAGE * STAGE}")
    e(f"s_mov_b32 {sr(S_TLEFT)}, %[ntiles]")
    e(f"s_mov_b32 {sr(S_KC)}, 0")
    e(f"s_mov_b32 {sr(S_CST)}, 0")
    e(f"s_mov_b32 {sr(S_RDELTA)}, {STAGE}")
    e(f"s_mov_b32 {sr(S_STG)}, {STAGE}")
    e(f"s_mov_b32 {sr(S_NSTG2)}, {-(NSTAGE - 1) * STAGE & 0xffffffff}")
    for i in range(16 * NB):
        e(f"v_accvgpr_write_b32 a{i}, 0")
    # first tile entry
    e(f"s_mov_b32 {sr(S_TLI)}, 0")
    for x in tile_entry(CUR):
        e(x)
    for i in range(4):
        e(f"s_mov_b32 {sr(NXT + i)}, {sr(CUR + i)}")
    # K tiles 0 and 1
    e(f"s_mov_b32 {sr(S_DDST)}, %[ldsb]")
    e(f"s_mul_i32 {sr(S_WDA)}, %[wave], {PA_N * 1024}")
    e(f"s_mul_i32 {sr(S_WDW)}, %[wave], {PW_N * 1024}")
    for kt in range(LOOK):
        for s_ in dma_dst_setup():
            e(s_)
        for m0w, cp in dma_pieces():
            e(m0w)
            e("s_nop 0")
            e(cp)
        for s in rotate_dma_dst():
            e(s)
        if kt < LOOK - 1:
            for p in (P_AH, P_AL, P_WH, P_WL):
                e(f"s_add_u32 {sr(p)}, {sr(p)}, {64 * KSUB}")
                e(f"s_addc_u32 {sr(p + 1)}, {sr(p + 1)}, 0")
    e(f"s_mov_b32 {sr(S_KD)}, {LOOK - 1}")
    if MX and EPI == "e2":
        import struct
        e(f"s_mov_b32 {sr(S_SCV)}, 0x{struct.unpack('<I', struct.pack('<f', 2.0 ** MX_EA))[0]:08x}")
        e(f"s_mov_b32 {sr(S_SCR)}, 0x{struct.unpack('<I', struct.pack('<f', 2.0 ** -MX_LA))[0]:08x}")
        e(f"s_mov_b32 {sr(S_PSEL)}, 0x05010400")
        e("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")     # FP16_OVFL: fp16 / fp8 conversions saturate (measured: mx_cvt_probe)
    e(f"s_waitcnt vmcnt({(LOOK - 1) * NP})")
    e("s_barrier")
    for k, i in x_order():
        e(read_ins(FX, k, i, 0))
    e("s_branch L_plain")


def tail():
    """the drain of the workgroup's last tile: the 16 steps in a row"""
    e("L_tail:")
    e("s_waitcnt vmcnt(0)")
    vm = []
    for d in range(NDRAIN):
        for it in drain_items(d):
            if isinstance(it, str):
                e(it)
            elif it[0] == "vm":
                e(it[1])
                vm.append(it[2])
            elif it[0] == "needvm":
                idx = max(i for i, nm in enumerate(vm) if nm == it[1])
                e(f"s_waitcnt vmcnt({min(len(vm) - idx - 1, 63)})")
            elif it[0] == "needprev":
                idx = [i for i, nm in enumerate(vm) if nm == f"Rstep{it[1]}"]
                if idx:
                    e(f"s_waitcnt vmcnt({min(len(vm) - max(idx) - 1, 63)})")
    e("s_waitcnt vmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")
    if MX and EPI == "e2":
        e("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0")


def lint(L):
    """manual hazards of gfx950 that nothing else checks for hand-written code (wait states: an instruction = 1, s_nop n = n + 1)"""
    def toks(ins):
        return ins.replace(",", " ").replace("|", " ").replace("-v", " v").split()

    def regs(tok):
        if len(tok) > 2 and tok[0] in "vsa" and tok[1] == "[":
            a, b = tok[2:-1].split(":")
            return {f"{tok[0]}{i}" for i in range(int(a), int(b) + 1)}
        return {tok}

    def srcs(ins):
        r = set()
        for t in toks(ins)[2:]:
            r |= regs(t)
        return r

    def dst(ins):
        return regs(toks(ins)[1])

    def within(i, n):
        """instructions following L[i] that start fewer than n wait states after it"""
        w, res = 0, []
        for x in L[i + 1:i + 12]:
            if x.endswith(":"):
                continue
            if w >= n:
                break
            res.append(x)
            w += int(x.split()[1]) + 1 if x.startswith("s_nop") else 1
        return res

    for i, ins in enumerate(L):
        op = ins.split()[0]
        if op.endswith(":") or op.startswith("s_waitcnt") or op in ("s_barrier", "s_nop", "s_branch") or op.startswith("s_cbranch"):
            continue
        if op in ("v_exp_f32", "v_rcp_f32"):                               # trans result -> VALU read: 1 wait state
            for x in within(i, 1):
                assert dst(ins).isdisjoint(srcs(x)), (i, ins, x)
        if op.startswith("v_cmp"):                                         # VALU mask write -> VALU mask read: 2 wait states
            for x in within(i, 2):
                assert not (x.startswith("v_cndmask") and not dst(ins).isdisjoint(regs(toks(x)[-1]))), (i, ins, x)
        if op.startswith("v_") and not op.startswith("v_cmp"):             # VALU VGPR write -> DPP read: 2 wait states
            for x in within(i, 2):
                if x.startswith("v_mov_b32_dpp"):
                    assert dst(ins).isdisjoint(regs(toks(x)[2])), (i, ins, x)
        if op.startswith("s_") and toks(ins)[1] == "m0":                   # SALU m0 write -> LDS-DMA: 1 wait state
            for x in within(i, 1):
                assert not x.startswith("global_load_lds"), (i, ins, x)
        if op == "v_readfirstlane_b32":                                    # VALU SGPR write -> SALU / VMEM read: 5 wait states (s_nop 4)
            for x in within(i, 5):
                if not x.startswith("v_readfirstlane"):
                    assert dst(ins).isdisjoint(srcs(x)), (i, ins, x)


def main_mx2():
    """bodies of the split form: `last` is generated first (the drain bodies count the VMEM ops behind ITS loads) and emitted last"""
    prologue()
    mark = len(out)
    e("L_last:")
    n_last, after_last = body_mx2("last")
    e(f"s_mov_b32 {sr(S_KC)}, 0")
    e(f"s_sub_u32 {sr(S_TLEFT)}, {sr(S_TLEFT)}, 1")
    e(f"s_cmp_eq_u32 {sr(S_TLEFT)}, 0")
    e("s_cbranch_scc0 L_drain")
    last_lines = out[mark:]
    del out[mark:]
    e("L_drain:")
    infos, prev = [], after_last
    for d in range(NDRAIN):
        nv, aft = body_mx2("drain", d, prev if prev is not None else 0)
        infos.append((nv, aft))
        prev = aft
        e(f"s_add_u32 {sr(S_KC)}, {sr(S_KC)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_KC)}, {sr(S_NKM1)}")
    e("s_cbranch_scc0 L_last")
    e("L_plain:")
    body_mx2("plain")
    e(f"s_add_u32 {sr(S_KC)}, {sr(S_KC)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_KC)}, {sr(S_NKM1)}")
    e("s_cbranch_scc1 L_plain")
    out.extend(last_lines)
    tail()
    return infos


def main():
    if MX and MX_SPLIT:
        infos = main_mx2()
        finish(infos)
        return
    prologue()
    e3on = EPI == "e3" and "nodrain" not in FLAGS
    nst = lambda d: n_stores(d) if (e3on and "nostore" not in FLAGS) else 0   # stores of e3 step d (issued behind the loads of d + 1)
    pl = (NP, 0)                                         # a plain body: its DMA pieces only (the minimum any body issues)
    nset = NBW + (n_loads(0) if e3on else 0)             # set-up loads of `last` (bias, residual values of step 0)
    last_info = (NP + nset, 0 if DMA_POST else nset)     # (DMA_POST: they are issued before its DMA pieces)
    after = lambda d: (NP if DMA_POST else 0) + (nst(d - 1) if d else 0)   # VMEM ops a body issues behind the residual loads of step d
    e("L_drain:")
    infos = []
    for d in range(NDRAIN):
        h1 = infos[d - 1] if d >= 1 else last_info
        h2 = infos[d - 2] if d >= 2 else (last_info if d == 1 else pl)
        infos.append(body("drain", d, (h1, h2), after(d)))
        e(f"s_add_u32 {sr(S_KC)}, {sr(S_KC)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_KC)}, {sr(S_NKM1)}")
    e("s_cbranch_scc0 L_last")
    e("L_plain:")
    body("plain", None, (pl, pl))
    e(f"s_add_u32 {sr(S_KC)}, {sr(S_KC)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_KC)}, {sr(S_NKM1)}")
    e("s_cbranch_scc1 L_plain")
    e("L_last:")
    li = body("last", None, (pl, pl))
    assert li == last_info or 'nodma' in FLAGS, (li, last_info)
    e(f"s_mov_b32 {sr(S_KC)}, 0")
    e(f"s_sub_u32 {sr(S_TLEFT)}, {sr(S_TLEFT)}, 1")
    e(f"s_cmp_eq_u32 {sr(S_TLEFT)}, 0")
    e("s_cbranch_scc0 L_drain")
    tail()
    finish(infos)


def finish(infos):
    lint(out)
    import re
    out[:] = [re.sub(r"\bL_\w+", lambda m: m.group(0) + "_%=", ln) for ln in out]
    CFGU = CFG.upper()
    nm = f"X4G_{CFGU}_{EPI.upper()}"
    clob = [f'"v{i}"' for i in range(32, 256)] + [f'"a{i}"' for i in range(0, 128 + 16 * NB)] + [f'"s{i}"' for i in list(range(8, 32)) + list(range(34, 96))] + ['"vcc"', '"scc"', '"memory"']
    txt = (f"// GENERATED by tools/gen/gen_gemm_x4g.py {CFG} {EPI} - do not edit.\n"
           f"#define X4G_{CFGU}_LDS_BYTES {NSTAGE * STAGE}\n#define X4G_{CFGU}_KTILE {32 * KSUB}\n#define X4G_{CFGU}_MIN_NK {NDRAIN + 1}\n"
           + (f"#define X4G_{CFGU}_SCALE_A {MX_SCALE_A}\n" if MX else "") +
           f"// {len(out)} instructions; workgroup tile {TM} x {TN}, LDS {NSTAGE * STAGE} bytes ({NSTAGE} stages)\n"
           f"#define {nm}_BODY \\\n" + " \\\n".join('    "' + ln + '\\n\\t"' for ln in out) + "\n"
           f"#define {nm}_CLOBBERS " + ", ".join(clob) + "\n")
    open(OUT, "w").write(txt)
    print(len(out), "instructions;", sum(1 for x in out if x.startswith("v_mfma")), "MFMAs;", NSTAGE, "stages; drain (ops, post)", infos)


if __name__ == "__main__":
    main()
