"""FETCH_SIZE / WRITE_SIZE (KB, rocprofv3 derived counters) of the bench's dominant kernels -> JSON for bench.py
(`traffic_from_committed_pmc`).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
a wide coalesced read stream (128-byte requests tallied at 64 B) => the read side is doubled; WRITE_SIZE is taken as is.
Per kernel the launches of the steady state are the longest ones (full 7-frame bank / full 16-object batch)."""
import json
import sqlite3
import sys

B, NK, TOK = 16, 28736, 4096
KERNELS = {   # key -> (name pattern, algorithmic bytes per launch)
    # Q fp32 + K hi plane + V^T planes (lo only for the 64 pointer tokens) read once, output planes written once
    # round 4 (assembly kernel): Q fragments fp16 + K plane + V^T plane read once, unnormalised rows + (max, sum) written once
    "cross_attention": ("%k_attention_x4a%", B * (2.0 * TOK * 256 + 2.0 * NK * 256 + 2.0 * NK * 64 + 4.0 * TOK * 64 + 8.0 * TOK)),
    # its query pass (fp32 queries in, fp16 fragments out) and its merge (rows + statistics in, two bf16 planes out)
    "cross_attention_qprep": ("%k_x4a_qprep%", B * TOK * 256 * (4.0 + 2.0)),
    "cross_attention_merge": ("%k_w8_merge<64>%", B * TOK * (64 * 4.0 + 8.0 + 64 * 4.0)),
    # the 8-wave kernel (mode bf16x3 / DS2_ATTN_X4A=0): Q fp32 + K hi plane + V^T planes (lo only for the 64 pointer tokens)
    "cross_attention_w8": ("%k_attention_w8<64%", B * (4.0 * TOK * 256 + 2.0 * NK * 256 + 2.0 * NK * 64 + 2.0 * 64 * 64 + 4.0 * TOK * 64)),
    # memory-attention FFN, fused: X planes in, residual in, result out (fp32), LN(result) out as two planes (3 of 4 layers; fp32 in
    # the last) + the weights once.  (FETCH_SIZE counts L2 misses: the 4 MB of weight planes every 128-row block re-reads compete
    # with the token streams for a 4 MB L2 per XCD and are partly served from the MALL, not from HBM.)
    # round 5, assembly GEMM at the 16-frame encoder batch (the longest launches of each form):
    # mlp.layers.1 of Hiera stage 3 (65536 x 576 x 2304, residual in place): A planes + W planes + residual in, result out
    "k_gemm_x4g_23_e3": ("%k_gemm_x4g_23_e3%", 65536.0 * 2304 * 4 + 576.0 * 2304 * 4 + 2 * 65536.0 * 576 * 4),
    # mlp.layers.0 (65536 x 2304 x 576, GELU): A planes + W planes in, result planes out
    "k_gemm_x4g_23_e2": ("%k_gemm_x4g_23_e2%", 65536.0 * 576 * 4 + 2304.0 * 576 * 4 + 65536.0 * 2304 * 4),
    # round 6, the MX form of the same layers (same plane bytes: fp16 + 2 x fp8 per element) and of attn.qkv (65536 x 1728 x 576: A planes +
    # W planes in, fp32 result out)
    "k_gemm_x4gm_23_e3": ("%k_gemm_x4gm_23_e3%", 65536.0 * 2304 * 4 + 576.0 * 2304 * 4 + 2 * 65536.0 * 576 * 4),
    "k_gemm_x4gm_23_e2": ("%k_gemm_x4gm_23_e2%", 65536.0 * 576 * 4 + 2304.0 * 576 * 4 + 65536.0 * 2304 * 4),
    "k_gemm_x4gm_23_e1": ("%k_gemm_x4gm_23_e1%", 65536.0 * 576 * 4 + 1728.0 * 576 * 4 + 65536.0 * 1728 * 4),
    # (round 5, LayerNorm inside: the un-normalised fp32 rows in - they are the residual as well, read twice -, result out, LN(result) planes out)
    "k_mlp256": ("%k_mlp256<1%", B * TOK * 256 * (4.0 + 4.0 + 4.0 + 4.0) + 2 * 2048 * 256 * 4.0),
    # round 5, fused kernels of a memory-attention layer (16 objects): planes in / fp32 q + fp16 k plane + V^T hi plane out; fp32 rows in /
    # fp16 fragments out; un-normalised rows + (max, sum) + residual in / residual stream out
    "k_qkv_self": ("%k_qkv_self%", B * TOK * 256 * (4.0 + 4.0 + 2.0 + 2.0) + 768 * 256 * 4.0),
    "k_qproj_x4a": ("%k_qproj_x4a%", B * TOK * 256 * (4.0 + 2.0) + 256 * 256 * 4.0),
    "k_vo_merge": ("%k_vo_merge%", B * TOK * (64 * 4.0 + 8.0 + 256 * 4.0 + 256 * 4.0) + 256 * 64 * 4.0),
}


def per_dispatch(db, counter, pattern):
    c = sqlite3.connect(db)
    return c.execute("select dispatch_id, sum(value), max(duration) from counters_collection where kernel_name like ? "
                     "and counter_name = ? group by dispatch_id", (pattern, counter)).fetchall()


def steady(rows):
    dmax = max(r[2] for r in rows)
    sel = [r[1] for r in rows if r[2] >= 0.9 * dmax]
    return sum(sel) / len(sel), len(sel)


out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, DS2_ASYNC_ENCODE=0) on "
                 "`python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream`; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024",
       "read_correction": 2.0}
for key, (pat, alg) in KERNELS.items():
    f, w = per_dispatch(sys.argv[1], "FETCH_SIZE", pat), per_dispatch(sys.argv[2], "WRITE_SIZE", pat)
    if not f or not w:
        continue
    fk, nf = steady(f)
    wk, nw = steady(w)
    out[key] = {"kernel": pat.strip("%"), "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                "launches_averaged": [nf, nw], "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "algorithmic_bytes": alg}
print(json.dumps(out, indent=1))
