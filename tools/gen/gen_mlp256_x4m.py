#!/usr/bin/env python
"""Generator of the hidden-dimension loop of the fused MLP (gemm_mlp256.hip) as ONE inline-assembly statement for gfx950:

    python tools/gen/gen_mlp256_x4m.py det-sam2_amd/csrc/mlp256_x4m_body.inc

Why: the C++ loop leaves the matrix pipe idle 60 % of the time (tools/mlp_time.py: 359 us per launch, of which 131 us are MFMAs and 68 us
the row block's prologue / epilogue; without LDS fragment reads 240 us, without the in-loop DMA 309, without barriers 336 - and halving
the reads changes nothing: it is their LATENCY, hipcc waits with lgkmcnt(0) one group ahead).  Here every fragment read is issued D units
(2 MFMAs each) ahead of its use and waited for with a COUNTED lgkmcnt, the DMA runs 6 steps ahead, the activation of a chunk runs as
fillers behind the MFMAs of the neighbouring steps.

Form (ReLU, two fp16 terms - the memory-attention FFN; everything else stays on the C++ loop).  Same arithmetic, same order per element
as the C++ loop (tools/mlp_layout_check.py compares the two builds bit for bit):
  * a chunk = 64 hidden units = 8 tiles of 16 KiB (64 rows x 128 B, hi plane at +0, lo plane at +8192): A0..A3 = W1 rows of the chunk x
    64 k each, B0..B3 = 64 output columns each x the chunk's 64 (permuted) hidden units.  Tile j of a chunk lives in ring slot j (8 slots).
  * a step = one tile = 8 units; unit = one fragment pair (hi, lo) = 2 MFMAs (lo first).  A tiles: unit (k-step, hidden block) into
    hid[hb] (VGPRs; the first MFMA of a chunk takes C = 0); B tiles: unit (k-step t, column block) into the caller's accumulators (AGPR
    operands o0..o7), B operand = the activated hidden fragment fh[t].
  * reads run D = 4 units ahead into a ring of 5 unit buffers.  The barrier that guards tile g + 1 sits in step g before the first read of
    tile g + 1 (unit 8 - D); behind it the wave issues the DMA of tile g + 7 into the slot of tile g - 1 - every wave has issued the last
    MFMA of tile g - 1 before it reached that barrier, so its fragments are in registers (strict, no reliance on latencies).
  * in-order queues simulated: lgkmcnt (ds_read only) per use, vmcnt(20) at every barrier (5 tiles x 4 pieces younger than the awaited one);
    the tail wraps the DMA to the first chunk so the accounting stays uniform (as the C++ loop does).
Operands: o0..o7 "+a" f32x16; x0..x15 "v" f16x8 (B operands, k-step s = x[s]); rd0, offa0, offa1, offb0, offb1, baddr "v"; w1h, w1l, w2h,
w2l (64-bit, chunk c_begin already applied), stridea, strideb (bytes: 64 W1 rows; 64 W2 rows), ldsb, wave, nch "s".
"""
import os
import sys

D = int(os.environ.get("X4M_D", "6"))          # read-ahead in units (measured: 4 .. 7 within 1 %)
NBUF = 8                                       # unit buffers: a divisor of the 64 units of a chunk (the loop body), > D
assert D < NBUF
MED3 = os.environ.get("X4M_MED3", "1") == "1"           # ReLU + saturation as one v_med3_f32
ILV = os.environ.get("X4M_ILV", "0") == "1"             # two units at a time: lo(U), lo(U+1), hi(U), hi(U+1) - no MFMA reads the accumulator the previous one wrote
WAIT1 = os.environ.get("X4M_WAIT1", "1") == "1"         # one lgkm wait per unit (for its hi fragment, the younger one) instead of two
NONOP = os.environ.get("X4M_NONOP", "1") == "1"         # the wait state between an m0 write and its LDS-DMA is filled by the unit's reads
FLAGS = set(os.environ.get("X4M_FLAGS", "").split())   # timing ablations (results WRONG): nobar noread nodma noact
ACT = os.environ.get("X4M_ACT", "relu")                 # relu (memory-attention FFN) | gelu (memory encoder CXBlock: common.h ds2_gelu)
assert ACT in ("relu", "gelu")
NTMP = 14 if ACT == "gelu" else 0                       # gelu: 3 temporaries x 4 elements in flight + the polynomial's constant term (+ 1 pad)
# ---- registers owned by the body (clobbered)
V0 = 256 - (32 + 16 + 32 + 8 * NBUF + 8 + 2 + NTMP)
assert V0 % 2 == 0
HID = V0                   # 32: hid[hb] = HID + 16 hb
FH = HID + 32              # 16: fh[i] = FH + 4 i, i = 2 hb + t
BIAS = FH + 16             # 32: bias[hb] = BIAS + 16 hb (4 x b128)
FR = BIAS + 32             # 40: unit buffer b: lo fragment FR + 8 b, hi fragment FR + 8 b + 4
RD = FR + 8 * NBUF         # 8: RD + k (slots 0..3), RD + 4 + k (slots 4..7), k = k-step
VB = RD + 8                # 1: bias read address of the current chunk
C65 = VB + 1               # 1: 65504.0
TMP = C65 + 1              # gelu: z, t, p of element i at TMP + 3 i, i = 0..3
C3 = TMP + 12              # gelu: -1.453152027
VEND = C65 + 1 + NTMP
assert VEND == 256, VEND
S0 = 36
PA_H, PA_L, PB_H, PB_L = S0, S0 + 2, S0 + 4, S0 + 6            # current chunk: W1 hi / lo, W2 hi / lo (64-bit)
NA_H, NA_L, NB_H, NB_L = S0 + 8, S0 + 10, S0 + 12, S0 + 14      # next chunk (wrapped to the first one behind the last)
BA_H, BA_L, BB_H, BB_L = S0 + 16, S0 + 18, S0 + 20, S0 + 22     # first chunk
T_H, T_L = S0 + 24, S0 + 26                                    # tile base of the DMA being issued
S_CNT, S_SA, S_DST, S_NEG = S0 + 28, S0 + 29, S0 + 30, S0 + 31
SB1, SB2, SB3 = S0 + 32, S0 + 33, S0 + 34                      # q * strideb
S_C1, S_C2 = S0 + 35, S0 + 36                                  # gelu: sqrt(1/2), 0.3275911
MSK = S0 + 38                                                  # gelu: 4 lane-mask pairs (x >= 0)
SEND = S0 + 46 if ACT == "gelu" else S0 + 35

L = []                     # emitted instructions


def emit(s):
    if "nobar" in FLAGS and s == "s_barrier":
        return
    if "noread" in FLAGS and (s.startswith("ds_read") or s.startswith("s_waitcnt lgkmcnt")):
        return
    if "nodma" in FLAGS and (s.startswith("global_load_lds") or s.startswith("s_waitcnt vmcnt")):
        return
    if "noact" in FLAGS and s.split()[0] in ("v_add_f32", "v_max_f32", "v_med3_f32"):
        return
    L.append(s)


class Queues:
    """in-order LGKM queue (ds_read only)"""

    def __init__(self):
        self.lgkm = []

    def read(self, tok):
        self.lgkm.append(tok)

    def wait(self, toks):
        """wait until every token of `toks` has returned"""
        idx = [i for i, t in enumerate(self.lgkm) if t in toks]
        if not idx:
            return
        n = min(len(self.lgkm) - 1 - max(idx), 15)                  # (the counter has 4 bits: waiting for more than needed is safe)
        emit(f"s_waitcnt lgkmcnt({n})")
        self.lgkm = self.lgkm[len(self.lgkm) - n:] if n else []


Q = Queues()


def unit_desc(j, u):
    """(kind, k, blk): A tile -> ('A', k-step, hidden block); B tile -> ('B', t, column block)"""
    if j < 4:
        if j == 3 and not ILV:
            return "A", u & 3, u >> 2          # hb-major: hid[0] is complete after unit 3
        return "A", u >> 1, u & 1
    return "B", u >> 1, u & 1


def read_unit(j, u, tag):
    """the two fragment reads of unit (j, u) -> buffer (unit index % NBUF); tag distinguishes this chunk's from the next chunk's units"""
    _, k, blk = unit_desc(j, u)
    b = (8 * j + u) % NBUF
    base = RD + k + (4 if j >= 4 else 0)
    off = (j & 3) * 16384 + blk * 4096
    emit(f"ds_read_b128 v[{FR + 8 * b}:{FR + 8 * b + 3}], v{base} offset:{off + 8192}")     # lo
    Q.read((tag, j, u, "lo"))
    emit(f"ds_read_b128 v[{FR + 8 * b + 4}:{FR + 8 * b + 7}], v{base} offset:{off}")         # hi
    Q.read((tag, j, u, "hi"))


def vreg4(r):
    return f"v[{r}:{r + 3}]"


def mfma(j, u, plane, first_of_chunk):
    kind, k, blk = unit_desc(j, u)
    b = (8 * j + u) % NBUF
    frag = vreg4(FR + 8 * b + (0 if plane == "lo" else 4))
    if kind == "A":
        acc = f"v[{HID + 16 * blk}:{HID + 16 * blk + 15}]"
        c = "0" if first_of_chunk else acc
        emit(f"v_mfma_f32_32x32x16_f16 {acc}, {frag}, %[x{4 * j + k}], {c}")
    else:
        o = f"%[o{2 * (j - 4) + blk}]"
        emit(f"v_mfma_f32_32x32x16_f16 {o}, {frag}, {vreg4(FH + 4 * k)}, {o}")


def dma_tile(j, nxt):
    """instruction groups (lists) issuing this wave's 4 pieces of tile j of the current (nxt = False) or the next chunk"""
    pa = (NA_H, NA_L, NB_H, NB_L) if nxt else (PA_H, PA_L, PB_H, PB_L)
    groups = []
    if j < 4:
        setup = [f"s_add_u32 s{T_H}, s{pa[0]}, {j * 128}", f"s_addc_u32 s{T_H + 1}, s{pa[0] + 1}, 0",
                 f"s_add_u32 s{T_L}, s{pa[1]}, {j * 128}", f"s_addc_u32 s{T_L + 1}, s{pa[1] + 1}, 0"]
        off = ("%[offa0]", "%[offa1]")
    else:
        q = j - 4
        if q == 0:
            setup = [f"s_mov_b64 s[{T_H}:{T_H + 1}], s[{pa[2]}:{pa[2] + 1}]", f"s_mov_b64 s[{T_L}:{T_L + 1}], s[{pa[3]}:{pa[3] + 1}]"]
        else:
            sb = (SB1, SB2, SB3)[q - 1]
            setup = [f"s_add_u32 s{T_H}, s{pa[2]}, s{sb}", f"s_addc_u32 s{T_H + 1}, s{pa[2] + 1}, 0",
                     f"s_add_u32 s{T_L}, s{pa[3]}, s{sb}", f"s_addc_u32 s{T_L + 1}, s{pa[3] + 1}, 0"]
        off = ("%[offb0]", "%[offb1]")
    first = True
    for plane, t in ((0, T_H), (8192, T_L)):
        for p in range(2):
            g = (setup if first else []) + [f"s_add_u32 m0, s{S_DST}, {j * 16384 + plane + p * 1024}", "s_nop 0",
                                            f"global_load_lds_dwordx4 {off[p]}, s[{t}:{t + 1}]"]

            first = False
            groups.append(g)
    return groups


def gelu_ops(hb, t):
    """bias + ds2_gelu + saturate + pack of hid[hb][8t .. 8t+7] -> fh[2 hb + t]: hipcc's instruction sequence for common.h ds2_gelu (contraction
    off, explicit fmaf), four elements at a time stage by stage - a transcendental's result is used 3 instructions later, a lane mask 7"""
    ops = []
    h, b, f = HID + 16 * hb + 8 * t, BIAS + 16 * hb + 8 * t, FH + 4 * (2 * hb + t)
    for e0 in (0, 4):
        x = [h + e0 + i for i in range(4)]
        bb = [b + e0 + i for i in range(4)]
        z, tt, pp = [TMP + 3 * i for i in range(4)], [TMP + 3 * i + 1 for i in range(4)], [TMP + 3 * i + 2 for i in range(4)]
        m = [f"s[{MSK + 2 * i}:{MSK + 2 * i + 1}]" for i in range(4)]
        stages = [
            lambda i: f"v_add_f32 v{x[i]}, v{x[i]}, v{bb[i]}",
            lambda i: f"v_mul_f32_e64 v{z[i]}, |v{x[i]}|, s{S_C1}",                 # z = |x| sqrt(1/2)
            lambda i: f"v_fma_f32 v{tt[i]}, v{z[i]}, s{S_C2}, 1.0",
            lambda i: f"v_rcp_f32 v{tt[i]}, v{tt[i]}",                             # t = 1 / (1 + 0.3275911 z)
            lambda i: f"v_fmamk_f32 v{pp[i]}, v{tt[i]}, 0x3f87dc22, v{C3}",         # p = t 1.061405429 - 1.453152027
            lambda i: f"v_fmaak_f32 v{pp[i]}, v{tt[i]}, v{pp[i]}, 0x3fb5f0e3",
            lambda i: f"v_fmaak_f32 v{pp[i]}, v{tt[i]}, v{pp[i]}, 0xbe91a98e",
            lambda i: f"v_fmaak_f32 v{pp[i]}, v{tt[i]}, v{pp[i]}, 0x3e827906",
            lambda i: f"v_mul_f32 v{pp[i]}, v{tt[i]}, v{pp[i]}",                    # poly
            lambda i: f"v_mul_f32 v{pp[i]}, 0.5, v{pp[i]}",
            lambda i: f"v_mul_f32_e64 v{z[i]}, v{z[i]}, -v{z[i]}",                  # -z^2
            lambda i: f"v_mul_f32 v{z[i]}, 0x3fb8aa3b, v{z[i]}",
            lambda i: f"v_exp_f32 v{z[i]}, v{z[i]}",
            lambda i: f"v_mul_f32 v{pp[i]}, v{z[i]}, v{pp[i]}",                     # erfc(z) / 2
            lambda i: f"v_cmp_le_f32_e64 {m[i]}, 0, v{x[i]}",
            lambda i: f"v_sub_f32 v{z[i]}, 1.0, v{pp[i]}",
            lambda i: f"v_cndmask_b32_e64 v{pp[i]}, v{pp[i]}, v{z[i]}, {m[i]}",
            lambda i: f"v_mul_f32 v{x[i]}, v{x[i]}, v{pp[i]}",
            lambda i: f"v_med3_f32 v{x[i]}, v{x[i]}, s{S_NEG}, v{C65}",
        ]
        for st in stages:
            for i in range(4):
                ops.append(st(i))
        for pr in range(2):
            ops.append(f"v_cvt_pk_f16_f32 v{f + e0 // 2 + pr}, v{h + e0 + 2 * pr}, v{h + e0 + 2 * pr + 1}")
    return ops


def act_ops(hb, t):
    """bias + ReLU + saturate + pack of hid[hb][8t .. 8t+7] -> fh[2 hb + t]; the instruction sequence hipcc emits for the C++ loop"""
    if ACT == "gelu":
        return gelu_ops(hb, t)
    ops = []
    h, b, f = HID + 16 * hb + 8 * t, BIAS + 16 * hb + 8 * t, FH + 4 * (2 * hb + t)
    for e in range(8):
        ops.append(f"v_add_f32 v{h + e}, v{h + e}, v{b + e}")
    if MED3:     # min(max(x, 0), 65504) in one instruction: equal to v_max_f32 0 + v_med3_f32 (-65504, 65504) for every input (NaN -> 0 both ways)
        for e in range(8):
            ops.append(f"v_med3_f32 v{h + e}, v{h + e}, 0, v{C65}")
    else:
        for e in range(8):
            ops.append(f"v_max_f32 v{h + e}, 0, v{h + e}")
        for e in range(8):
            ops.append(f"v_med3_f32 v{h + e}, v{h + e}, s{S_NEG}, v{C65}")
    for p in range(4):
        ops.append(f"v_cvt_pk_f16_f32 v{f + p}, v{h + 2 * p}, v{h + 2 * p + 1}")
    return ops


def bias_reads():
    ops = []
    for hb in range(2):
        for g in range(4):
            r = BIAS + 16 * hb + 4 * g
            ops.append((f"ds_read_b128 v[{r}:{r + 3}], v{VB} offset:{(hb * 32 + 8 * g) * 4}", ("bias", hb, g)))
    return ops


def prologue():
    emit("s_waitcnt lgkmcnt(0)")                                  # (LDS writes of the C++ prologue: b1s / lns)
    emit(f"s_mov_b64 s[{PA_H}:{PA_H + 1}], %[w1h]")
    emit(f"s_mov_b64 s[{PA_L}:{PA_L + 1}], %[w1l]")
    emit(f"s_mov_b64 s[{PB_H}:{PB_H + 1}], %[w2h]")
    emit(f"s_mov_b64 s[{PB_L}:{PB_L + 1}], %[w2l]")
    for d, s in ((BA_H, PA_H), (BA_L, PA_L), (BB_H, PB_H), (BB_L, PB_L)):
        emit(f"s_mov_b64 s[{d}:{d + 1}], s[{s}:{s + 1}]")
    emit(f"s_mov_b32 s{S_CNT}, %[nch]")
    emit(f"s_mov_b32 s{S_SA}, %[stridea]")
    emit(f"s_mov_b32 s{SB1}, %[strideb]")
    emit(f"s_lshl_b32 s{SB2}, s{SB1}, 1")
    emit(f"s_add_u32 s{SB3}, s{SB2}, s{SB1}")
    emit(f"s_lshl_b32 s{S_DST}, %[wave], 11")                     # this wave's pieces: 2 wave, 2 wave + 1 (1 KiB each)
    emit(f"s_add_u32 s{S_DST}, s{S_DST}, %[ldsb]")
    emit(f"s_mov_b32 s{S_NEG}, 0xc77fe000")                       # -65504.0
    emit(f"v_mov_b32 v{C65}, 0x477fe000")                         # 65504.0
    if ACT == "gelu":
        emit(f"s_mov_b32 s{S_C1}, 0x3f3504f3")                    # sqrt(1/2)
        emit(f"s_mov_b32 s{S_C2}, 0x3ea7ba05")                    # 0.3275911
        emit(f"v_mov_b32 v{C3}, 0xbfba00e3")                      # -1.453152027
    emit(f"v_mov_b32 v{VB}, %[baddr]")
    for k in range(4):
        if k == 0:
            emit(f"v_mov_b32 v{RD}, %[rd0]")
        else:
            emit(f"v_xor_b32 v{RD + k}, {32 * k}, %[rd0]")
    for k in range(4):
        emit(f"v_add_u32 v{RD + 4 + k}, 0x10000, v{RD + k}")
    next_pointers(first=True)
    # tiles 0 .. 6 of the first chunk; the steps issue tile g + 7
    for j in range(7):
        for g in dma_tile(j, False):
            for ins in g:
                emit(ins)
    emit("s_waitcnt vmcnt(24)")
    emit("s_barrier")
    for u in range(D):
        read_unit(0, u, "cur")


def next_pointers(first=False):
    """next = (chunks left > 1) ? current + stride : first chunk.  S_CNT = chunks left including the current one"""
    emit(f"s_cmp_gt_u32 s{S_CNT}, 1")
    for n, p, b, st in ((NA_H, PA_H, BA_H, f"s{S_SA}"), (NA_L, PA_L, BA_L, f"s{S_SA}"), (NB_H, PB_H, BB_H, "128"), (NB_L, PB_L, BB_L, "128")):
        emit(f"s_add_u32 s{T_H}, s{p}, {st}")
        emit(f"s_addc_u32 s{T_H + 1}, s{p + 1}, 0")
        emit(f"s_cmp_gt_u32 s{S_CNT}, 1")                           # (s_addc clobbers SCC)
        emit(f"s_cselect_b64 s[{n}:{n + 1}], s[{T_H}:{T_H + 1}], s[{b}:{b + 1}]")


def chunk_body():
    emit("L_chunk_%=:")
    hiq = []             # activation VALU ops: (instruction, fh index it contributes to)
    loq = []             # groups with slack: ("dma", [instructions]) | ("read", instruction, token)

    def flush_lo():
        m = loq.pop(0)
        if m[0] == "read":
            emit(m[1])
            Q.read(m[2])
        else:
            for ins in m[1]:
                emit(ins)

    def flush_hi(n):
        k = 0
        while hiq and k < n:
            emit(hiq.pop(0)[0])
            k += 1
        return k

    def barrier_and_dma(j):
        while any(m[0] == "dma" for m in loq):                     # (the previous step's DMA belongs in front of this barrier's count)
            flush_lo()
        emit("s_waitcnt vmcnt(20)")
        emit("s_barrier")
        # DMA of tile g + 7: tile 7 of this chunk at j = 0, tile j - 1 of the next chunk otherwise
        for g in (dma_tile(7, False) if j == 0 else dma_tile(j - 1, True)):
            loq.append(("dma", g))

    def need_fh(k):
        """the activated fragment fh[k] must be complete before a B unit reads it"""
        forced = 0
        while any(0 <= tag <= k or (tag == -1 and any(t2 <= k for _, t2 in hiq if t2 >= 0)) for _, tag in hiq[:1]) or any(0 <= tag <= k for _, tag in hiq):
            forced += flush_hi(1)
        if forced or any(x.startswith("v_cvt_pk_f16_f32") for x in L[-2:]):
            emit("s_nop 1")                                        # VALU write -> MFMA operand read

    def bias_ready():
        while any(m[0] == "read" for m in loq):
            flush_lo()
        Q.wait({("bias", hb, g) for hb in range(2) for g in range(4)})

    def dma_pre():
        if NONOP and loq and loq[0][0] == "dma":
            g = loq.pop(0)[1]
            assert g[-2] == "s_nop 0"
            for ins in g[:-2]:
                emit(ins)
            return g[-1]
        return None

    def reads_ahead(j, u):
        jj, uu = j + (u + D) // 8, (u + D) % 8
        read_unit(jj % 8, uu, "cur" if jj < 8 else "nxt")

    for j in range(8):
        if ILV:
            assert D % 2 == 0
            for u in range(0, 8, 2):
                if u == 8 - D:
                    barrier_and_dma(j)
                if j == 1 and u == 0:
                    for ins, tok in bias_reads():
                        loq.append(("read", ins, tok))
                if (j, u) == (3, 6):
                    bias_ready()
                if (j, u) == (4, 0):                               # both hidden blocks are complete with the last pair of A3
                    hiq.append(("s_nop 15", -1))
                    for hb in range(2):
                        for t in range(2):
                            hiq.extend((ins, 2 * hb + t) for ins in act_ops(hb, t))
                kind, k, _ = unit_desc(j, u)
                if kind == "B":
                    need_fh(k)
                Q.wait({("cur", j, u + 1, "hi")})
                mfma(j, u, "lo", j == 0 and k == 0)
                load = dma_pre()
                reads_ahead(j, u)
                mfma(j, u + 1, "lo", j == 0 and k == 0)
                reads_ahead(j, u + 1)
                mfma(j, u, "hi", False)
                if load:
                    emit(load)
                load2 = dma_pre()
                if hiq:
                    flush_hi(6)
                mfma(j, u + 1, "hi", False)
                if load2:
                    emit(load2)
                if hiq:
                    flush_hi(8)
                elif loq and not load2:
                    flush_lo()
            continue
        for u in range(8):
            if u == 8 - D:                                        # tile j + 1 is read from here on: all of its pieces must have landed
                barrier_and_dma(j)
            if j == 1 and u == 0:
                for ins, tok in bias_reads():
                    loq.append(("read", ins, tok))
            if (j, u) in ((3, 5), (4, 1)):                         # hid[hb] is complete 2 units (4 MFMAs) earlier
                hb = 0 if j == 3 else 1
                hiq.append(("s_nop 3", -1))
                for t in range(2):
                    hiq.extend((ins, 2 * hb + t) for ins in act_ops(hb, t))
            kind, k, blk = unit_desc(j, u)
            if kind == "B":
                need_fh(k)
            if j == 3 and u == 4:                                  # the bias of both hidden blocks in registers before the activation starts
                bias_ready()
            Q.wait({("cur", j, u, "hi" if WAIT1 else "lo")})
            mfma(j, u, "lo", j == 0 and k == 0)
            # first slot: the m0 write (+ tile base) of the next DMA piece, then the reads of the unit D ahead (the next chunk's tile 0
            # behind step 7: tagged "nxt") - they fill the wait state between the m0 write and the LDS-DMA that follows the second MFMA
            load = dma_pre()
            reads_ahead(j, u)
            if not NONOP and hiq and loq:                          # the second slot belongs to the activation: keep the DMA moving here
                flush_lo()
            if not WAIT1:
                Q.wait({("cur", j, u, "hi")})
            mfma(j, u, "hi", False)
            if load:
                emit(load)
            if hiq:
                flush_hi(9)
            elif loq and not load:
                flush_lo()
    assert not hiq and not loq, (hiq, loq)
    # loop control: chunks left, pointers, bias address
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    for p, n in ((PA_H, NA_H), (PA_L, NA_L), (PB_H, NB_H), (PB_L, NB_L)):
        emit(f"s_mov_b64 s[{p}:{p + 1}], s[{n}:{n + 1}]")
    next_pointers()
    emit(f"v_add_u32 v{VB}, 256, v{VB}")
    emit(f"s_cmp_gt_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_chunk_%=")
    check_cycle()
    emit("s_waitcnt lgkmcnt(0)")


def check_cycle():
    """the LGKM queue at the end of the loop body = the one at its start (units 0 .. D-1 of the following tile 0), modulo the tag"""
    want = [("nxt", 0, u, p) for u in range(D) for p in ("lo", "hi")]
    assert Q.lgkm == want, Q.lgkm


def lint():
    for i, ins in enumerate(L):
        if ins.startswith("s_add_u32 m0") and "nodma" not in FLAGS:
            # SALU m0 write -> LDS-DMA: 1 wait state; and the next LDS-DMA is this write's, with no other m0 write before it
            assert not L[i + 1].startswith("global_load_lds"), (i, ins)
            nxt = next(k for k in range(i + 1, len(L)) if L[k].startswith("global_load_lds") or L[k].startswith("s_add_u32 m0"))
            assert L[nxt].startswith("global_load_lds"), (i, ins, L[nxt])
            assert not any(x.startswith("s_barrier") or x.endswith(":") or x.startswith("s_cbranch") for x in L[i:nxt]), (i, ins)
    n = sum(1 for x in L if x.startswith("v_mfma"))
    assert n == 128, n


def main():
    out = sys.argv[1]
    prologue()
    start = list(Q.lgkm)
    assert start == [("cur", 0, u, p) for u in range(D) for p in ("lo", "hi")]
    n0 = len(L)
    chunk_body()
    lint()
    # (the final s_waitcnt lgkmcnt(0) drained the queue; check the state just before it)
    body = L[n0:]
    clob = [f"v{i}" for i in range(V0, 256)] + [f"s{i}" for i in range(S0, SEND)] + ["m0", "scc", "memory"]
    with open(out, "w") as f:
        f.write("// generated by tools/gen/gen_mlp256_x4m.py - do not edit\n")
        f.write(f"#define {'X4M_GELU' if ACT == 'gelu' else 'X4M'}_V0 {V0}\n")
        tag = "X4M_GELU" if ACT == "gelu" else "X4M"
        f.write(f"#define {tag}_BODY \\\n")
        for ins in L:
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write(f"#define {tag}_CLOBBERS " + ", ".join(f'"{c}"' for c in clob if c != "m0") + "\n")
    print(out, len(L), "instructions;", sum(1 for x in body if x.startswith("s_waitcnt lgkmcnt")), "lgkm waits in the chunk body")


if __name__ == "__main__":
    main()
