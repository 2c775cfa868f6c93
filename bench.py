#!/usr/bin/env python
"""bench.py - frames/sec/GPU of propagate_in_video on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): sam2.1_hiera_large, 16 objects, 1024x1024 synthetic frames, 7-frame memory
bank (1 conditioning + 6 non-conditioning frames, 16 object pointers => Nk = 28736), synthetic
deterministic checkpoint (no SAM 2.1 weights exist offline).  A "step" = one tracked frame of
propagate_in_video: image encoder on a NEW frame (no feature-cache hit), bank gather, 4-layer memory
attention, SAM heads (multimask), memory encoder, 256->1024 upsample + threshold + bit-pack, and the packed
masks copied to the host.  Frames are resident in HBM (as in the reference, where init_state loads them
before propagate).  N>1: passes shard over ranks (weak scaling), with the RCCL all-gather of the cond-frame
bank entry inside the timed region.

Prints ONE JSON line on rank 0 (see the driver contract in the task description).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREFILL = 15  # tracked frames needed before the 7-frame bank + 16 pointers are in steady state
# extra frames tracked AFTER the timed region with one HIP-event bracket per GEMM (roofline_gemm): one whole encoder batch
GEMM_PROBE = int(os.environ.get("DS2_ENCODE_BATCH", "16"))
# MI355X_MICROARCH.md dense MFMA peaks: fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_*_bf16)
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16x3k": 2500.0}
DTYPE = {"fp32": "f32",
         "bf16x3": "bf16x3 (fp32 operands split into 2 bf16 planes, 3 bf16 MFMAs per product, fp32 accumulate/softmax/storage)",
         "bf16x3k": "bf16x3k (fp32 operands split into 2 bf16 planes, 3 bf16 MFMAs per product; in the memory attention the q / k / "
                    "softmax-weight / value operands are ONE fp16 plane each, the fused MLPs of the memory attention and memory "
                    "encoder use 2 fp16 MFMAs per product (activations one fp16 plane, weights two), and the Linear layers of Hiera stages "
                    "3 / 4 at widths that are multiples of 192 (hiera_l) use the two-MFMA-equivalent 'MX' product - fp16 hi.hi + both cross "
                    "terms in scaled fp8 MFMAs (e4m3, static power-of-two scales); fp32 accumulate/softmax/storage)"}


def cross_attention_flops(B, Nk, tokens=4096, d=256, dv=64):
    """Algorithmic FLOPs of ONE cross-attention launch (one layer, all B objects): QK^T over 256-d keys plus
    P.V in the 64-d memory space (DESIGN.md section 'Kernels')."""
    return 2.0 * B * tokens * Nk * (d + dv)


def cross_attention_bytes(B, Nk, tokens=4096, d=256, dv=64, precision="bf16x3", n_ptr_tok=64):
    """Algorithmic HBM bytes of one cross-attention launch: Q fp32 + K planes + V^T planes read once, output planes
    written once.  bf16x3: K and V^T as two bf16 planes each; bf16x3k: K as one plane, and the V^T lo plane only for the
    pointer tokens (the frame tokens are bf16 storage: their lo plane is zero and is not staged)."""
    if precision == "bf16x3k":
        return B * (4.0 * tokens * d + 2.0 * Nk * d + 2.0 * Nk * dv + 2.0 * n_ptr_tok * dv + 4.0 * tokens * dv)
    return 4.0 * B * (tokens * d + Nk * d + Nk * dv + tokens * dv)


def committed_pmc_traffic(kernel_key, B, nk, precision):
    """HBM bytes per launch of a kernel from the COMMITTED PMC passes (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs of this same bench command - counters cannot be collected from inside the timed run, so
    the bench line's own `traffic` is null and this number is reported under `traffic_from_committed_pmc` with its
    source file).  None if no committed file matches this workload."""
    if B != 16 or nk != 28736 or precision != "bf16x3k":
        return None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_cross_attention.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        if kernel_key in d:
            return {"bytes_per_launch": float(d[kernel_key]["traffic_bytes_per_launch"]), "source": f"profiles/{name}",
                    # (the PMC file prices the MAIN kernel of the launch only - k_attention_x4a without its query pre-pass and merge)
                    "algorithmic_bytes_of_that_kernel": d[kernel_key].get("algorithmic_bytes")}
        if kernel_key == "cross_attention" and "traffic_bytes_per_launch" in d:
            return {"bytes_per_launch": float(d["traffic_bytes_per_launch"]), "source": f"profiles/{name}"}
    return None


# algorithmic GFLOP of one tracked frame (2 x MAC; matmul + conv + SDPA as torch's FlopCounterMode
# counts them on the reference): F = F_enc + B * (F_ma(Nk) + 3.64 + 11.61), F_ma(Nk) = 116.0 + 0.017039 * Nk
F_ENC_GFLOP = {"sam2.1_hiera_t": 207.0, "sam2.1_hiera_b+": 529.3, "sam2.1_hiera_l": 1619.7}


def path_flops(model_name, B, nk):
    if model_name not in F_ENC_GFLOP:
        return None
    return (F_ENC_GFLOP[model_name] + B * (116.0 + 0.017039 * nk + 3.64 + 11.61)) * 1e9


def hiera_attention_flops(cfg):
    """Algorithmic FLOPs of the attention cores (QK^T + P.V, 2 x MAC) of ONE image through the Hiera trunk
    (sam2/modeling/backbones/hieradet.py:40-82 MultiScaleAttention inside :86-168 MultiScaleBlock; window partition with zero
    padding backbones/utils.py:16-96 - padded tokens are real keys and queries there, so they are counted).  Per block:
    4 * Lq_total * Lk_per_query * dim_out with Lk = window^2 (or all tokens for the global blocks) and the queries pooled
    2 x 2 inside the window where the block has a q_stride."""
    side = cfg.image_size // 4
    total = 0.0
    for b in cfg.trunk.blocks():
        w = b["window"]
        if w:
            pad = -(-side // w) * w
            nwin, lk = (pad // w) ** 2, w * w
            lq = lk // (b["q_stride"] ** 2) if b["q_stride"] else lk
            total += 4.0 * nwin * lq * lk * b["dim_out"]
        else:
            lk = side * side
            lq = lk // (b["q_stride"] ** 2) if b["q_stride"] else lk
            total += 4.0 * lq * lk * b["dim_out"]
        if b["q_stride"]:
            side //= b["q_stride"]
    return total


def metric_name(model_name, n_obj):
    """BASELINE.json's metric string for the headline workload; any other --model / --objects is named as what it is."""
    short = model_name.replace("sam2.1_", "").replace("hiera_large", "hiera_l").replace("hiera_base_plus", "hiera_b+")
    return f"frames/sec/GPU propagate_in_video, {short}, {n_obj} obj, 1024^2; mask IoU vs ref"


def cpu_baseline(model_name, n_obj=16):
    """Oracle (CPU restatement, oracle/) timed on the host cores on a bounded sample of the same workload: ONE tracked frame
    at the FULL object count (encoder + bank of 7 frames / 16 pointers + memory attention + SAM heads + memory encoder for
    `n_obj` objects in one batch, as the reference runs it) - measured, not scaled.  The one-object
    timing and the value the earlier rounds modelled from it (t_enc + n_obj * t_1) are reported beside it."""
    from det_sam2_amd.config import resolve_config
    from det_sam2_amd.synth import synthetic_frame
    from det_sam2_amd.weights import synthetic_state_dict
    from oracle import modeling as M
    from oracle.predictor import OraclePredictor, load_frames

    cfg = resolve_config(model_name)
    sd = synthetic_state_dict(cfg, 0)
    op = OraclePredictor(sd, cfg)
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    imgs, _, _ = load_frames([synthetic_frame(0)])

    def bank(B):
        mk = lambda: {"maskmem_features": torch.randn(B, 64, 64, 64, generator=g).to(torch.bfloat16),  # noqa: E731
                      "maskmem_pos_enc": [M.sine_pos_2d(64, 64, 64)[None].expand(B, -1, -1, -1)], "obj_ptr": torch.randn(B, 256, generator=g)}
        return {"cond_frame_outputs": {0: mk()}, "non_cond_frame_outputs": {t: mk() for t in range(1, 16)}}

    with torch.inference_mode():
        t0 = time.time()
        fpn, pos = M.forward_image(sd, cfg, imgs[0].float().unsqueeze(0))
        t_enc = time.time() - t0
        feats = [f.flatten(2).permute(2, 0, 1) for f in fpn]
        poss = [p.flatten(2).permute(2, 0, 1) for p in pos]
        t0 = time.time()
        op.track_step(16, False, feats, poss, None, None, bank(1), 64, False, True)
        t_one = time.time() - t0
        featsB = [f.expand(-1, n_obj, -1) for f in feats]
        possB = [p.expand(-1, n_obj, -1) for p in poss]
        t0 = time.time()
        op.track_step(16, False, featsB, possB, None, None, bank(n_obj), 64, False, True)
        t_all = time.time() - t0
    return {"value": 1.0 / (t_enc + t_all), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle (fp32 PyTorch-CPU restatement) on 1 tracked frame of {model_name} with {n_obj} objects in one batch, "
                      f"Nk=28736: encoder {t_enc:.1f}s + tracking {t_all:.1f}s (measured); one object alone {t_one:.1f}s",
            "modelled_from_one_object": 1.0 / (t_enc + n_obj * t_one)}


SUSTAINED_MFMA_TFLOPS = 1670.0   # measured: tools/experiments/x4a_bench.hip `strip` on all 256 CUs (profiles/r04_mfma_power_calibration.txt)


def cross_kernel_name(precision):
    if precision == "fp32":
        return "k_attention<256,64>"
    if precision == "bf16x3k" and os.environ.get("DS2_ATTN_X4A", "1") != "0":
        return "k_attention_x4a"
    return "k_attention_w8<64,2>"


def kernel_probe(pred, gen, st, last_tracked, table_path=None):
    """GEMM_PROBE more tracked frames of the same generator with HIP-event brackets per GEMM shape ("gemm M N K": the GEMM
    incl. its operand-split pre-pass), per GEMM KERNEL ("kern <name> M N K": the kernel alone) and per attention kernel,
    outside the timed region so the brackets cannot perturb `value` (ds2_profile_enable(2)).  The async encoder is
    switched off for these frames, so every duration is the kernel alone on the chip.
    -> (roofline_gemm: the family + its largest-time shape, by_kernel: {kernel: {ms_per_frame, flops_per_frame, launches}})."""
    for t in [t for t in st["cached_features"] if t > last_tracked]:     # the probe encodes exactly one batch of new frames
        st["cached_features"].pop(t)
    for t in [t for t in (st.get("_pending_features") or {}) if t > last_tracked]:
        st["_pending_features"].pop(t)
    was_async, pred.async_encode = pred.async_encode, False
    torch.cuda.synchronize()
    enc_before = pred.stats["encoder_runs"]
    pred.trace = []
    for tag in pred.hip.profile_tags():       # records the timed region left behind (e.g. kernel.hiera_attention) are not the probe's
        pred.hip.profile_read(tag)
    pred.hip.profile_enable(True, gemm_shapes=True)
    for _ in range(GEMM_PROBE):
        next(gen)
    torch.cuda.synchronize()
    pred.hip.profile_enable(False)
    pred.async_encode = was_async
    enc_runs = pred.stats["encoder_runs"] - enc_before
    B = int(st["output_dict"]["cond_frame_outputs"][0]["obj_ptr"].shape[0])
    nks = [tr["nk"] for tr in pred.trace]
    rows, kern = [], {}
    for tag in pred.hip.profile_tags():
        ms, n = pred.hip.profile_read(tag)
        if not n:
            continue
        if tag.startswith("gemm "):
            M, N, Kd = (int(x) for x in tag.split()[1:4])
            rows.append({"M": M, "N": N, "K": Kd, "calls_per_frame": n / GEMM_PROBE, "ms_per_frame": ms / GEMM_PROBE,
                         "avg_us": ms / n * 1e3, "tflops": 2.0 * M * N * Kd / (ms / n * 1e-3) / 1e12})
        elif tag.startswith("kern "):
            name, M, N, Kd = tag.split()[1], *(int(x) for x in tag.split()[2:5])
            k = kern.setdefault(name, {"ms": 0.0, "flops": 0.0, "launches": 0, "shapes": {}})
            k["ms"] += ms
            k["flops"] += 2.0 * M * N * Kd * n          # Kd is K rounded up to 32 (the pad columns are multiplied too)
            k["launches"] += n
            k["shapes"][(M, N, Kd)] = (ms, n)
        elif tag == "kernel.cross_attention":
            kern[cross_kernel_name(pred.hip.get_precision()) + " (memory cross-attention)"] = {
                "ms": ms, "flops": sum(cross_attention_flops(B, nk) for nk in nks) * pred.cfg.mem_attn_layers, "launches": n}
        elif tag == "kernel.self_attention":
            kern["k_attention_w8<256,1> (memory self-attention, incl. its V^T split)"] = {
                "ms": ms, "flops": len(nks) * (1 + (pred.cfg.mem_attn_layers - 1) * B) * 2.0 * 4096 * 4096 * 512, "launches": n}
        elif tag == "kernel.hiera_attention":
            # one encoder batch of GEMM_PROBE images ran inside the probe (asserted by the caller through encoder_runs)
            kern["Hiera attention (k_attn_winlds / k_attention_hg / k_attn_smallwin / k_attention_bf16x3)"] = {
                "ms": ms, "flops": hiera_attention_flops(pred.cfg) * enc_runs, "launches": n}
    if not rows:
        return None, {}
    rows.sort(key=lambda r: -r["ms_per_frame"])
    total = sum(r["ms_per_frame"] for r in rows)
    flops = sum(2.0 * r["M"] * r["N"] * r["K"] * r["calls_per_frame"] for r in rows)
    by_kernel = {}
    for name, k in kern.items():
        by_kernel[name] = {"ms_per_frame": k["ms"] / GEMM_PROBE, "launches_per_frame": k["launches"] / GEMM_PROBE,
                           "avg_launch_ms": k["ms"] / k["launches"],
                           "tflops": None if k["flops"] is None else k["flops"] / (k["ms"] * 1e-3) / 1e12,
                           "flops_per_launch": None if k["flops"] is None else k["flops"] / k["launches"]}
    if table_path:
        with open(table_path, "w") as f:
            f.write(f"# per-shape GEMM table, HIP events, {GEMM_PROBE} tracked frames, async encoder off; total {total:.3f} ms/frame, "
                    f"{flops / (total * 1e-3) / 1e12:.1f} algorithmic TFLOP/s overall\n")
            f.write(f"{'M':>8s} {'N':>6s} {'K':>6s} {'calls/frame':>12s} {'ms/frame':>9s} {'avg_us':>9s} {'TFLOP/s':>8s}\n")
            for r in rows:
                f.write(f"{r['M']:8d} {r['N']:6d} {r['K']:6d} {r['calls_per_frame']:12.2f} {r['ms_per_frame']:9.3f} {r['avg_us']:9.1f} {r['tflops']:8.1f}\n")
            f.write("# per kernel (the kernel alone, K rounded up to 32)\n")
            for name, k in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms_per_frame"]):
                f.write(f"# {name:70s} {k['ms_per_frame']:8.3f} ms/frame {k['launches_per_frame']:7.1f} launches/frame "
                        f"{(k['tflops'] or 0.0):8.1f} TFLOP/s\n")
                for (M, N, Kd), (ms, n) in sorted(kern[name].get("shapes", {}).items(), key=lambda kv: -kv[1][0]):
                    f.write(f"#     {M:8d} {N:6d} {Kd:6d} {n / GEMM_PROBE:8.2f}/frame {ms / n * 1e3:9.1f} us {2.0 * M * N * Kd / (ms / n * 1e-3) / 1e12:8.1f} TFLOP/s\n")
    top = rows[0]
    peak = PEAK_TFLOPS[pred.hip.get_precision()]
    fam = {"bound": "mfma", "kernel": "bf16x3 GEMM family (k_gemm_split*), largest-time shape", "shape": [top["M"], top["N"], top["K"]],
           "achieved": top["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": top["tflops"] / peak, "avg_launch_ms": top["avg_us"] * 1e-3,
           "calls_per_frame": top["calls_per_frame"], "family_ms_per_frame": total, "family_tflops": flops / (total * 1e-3) / 1e12,
           "family_frac": flops / (total * 1e-3) / 1e12 / peak,
           # the fused launches are GEMM work that left the per-shape table: k_mlp256 (memory-attention FFN, CXBlock; with the LayerNorm in
           # front of it since round 5), k_qkv_self (in_proj + key / value operand passes), k_qproj_x4a (norm2 + q_proj + query pass)
           "family_incl_fused_mlp_ms_per_frame": total + sum(v["ms"] for n_, v in kern.items() if n_.startswith(FUSED_GEMM_KERNELS)) / GEMM_PROBE,
           "family_incl_fused_mlp_tflops": (flops * GEMM_PROBE + sum(v["flops"] for n_, v in kern.items() if n_.startswith(FUSED_GEMM_KERNELS))) /
                                           ((total * GEMM_PROBE + sum(v["ms"] for n_, v in kern.items() if n_.startswith(FUSED_GEMM_KERNELS))) * 1e-3) / 1e12,
           "family_incl_fused_kernels": list(FUSED_GEMM_KERNELS),
           "note": "HIP-event bracket per GEMM (incl. its operand-split pre-pass when the producer did not emit planes), "
                   f"{GEMM_PROBE} frames after the timed region with the async encoder off; algorithmic FLOPs 2*M*N*K"}
    return fam, by_kernel


FUSED_GEMM_KERNELS = ("k_mlp256", "k_qkv_self", "k_qproj_x4a", "k_vo_merge")


def stream_fps(pred, B, n_frames):
    """Stream-level rate of VideoProcessor.run: Det-SAM2 defaults 30/30/60/60, B objects, HOST uint8 frames
    (H2D copy + ds2_ingest_frames + detections->prompts + second-visit tracking + eviction + packed masks to host all
    inside).  Every stream frame is tracked twice, so stream fps ~ tracked fps / 2."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
    frames = [synthetic_frame(t, 777) for t in range(n_frames)]
    vp = VideoProcessor(model_cfg=pred.cfg.name, detector=SyntheticDetector(B), skip_classes=set(), predictor=pred)
    t0_tracked = pred.stats["tracked_frames"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    segs = vp.run(frames=frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert sorted(segs) == list(range(n_frames)) and all(len(s) == B for s in segs.values())
    tracked = pred.stats["tracked_frames"] - t0_tracked
    return {"stream_fps": n_frames / dt, "frames": n_frames, "seconds": dt, "tracked_frames": tracked,
            "tracked_fps": tracked / dt, "schedule": "frame_buffer 30 / detect 30 / track 60 / keep 60 (Det-SAM2 defaults)",
            "input": "host uint8 RGB 1024x1024 frames; H2D + ingest + prompts + eviction + D2H of packed masks included"}


def bench_sharded(a, pred, cfg, world, rank, dev):
    """N > 1 (default): ONE Det-SAM2 stream whose propagate passes are sharded over the ranks (BASELINE config 4,
    det_sam2_amd.parallel.ShardedVideoProcessor).  The reference's 30/30/60/60 schedule scaled so that one pass tracks
    exactly K frames: frame_buffer = detect_interval = K/2, max_frame_num_to_track = max_inference_state_frames = K.
    Untimed: round 0 (passes 0..N-1, bank fill-up) [+ more rounds until W tracked frames per rank have run]; timed: ONE
    round = every rank ingests the round's N*K/2 host frames, encodes its own buffer, hands the pyramids to its ring
    neighbour (RCCL send/recv), all-gathers detections and the new conditioning entries (RCCL all-gather), and tracks its
    pass of K frames; masks packed and copied to the host.  value = N*K tracked frames / max-over-ranks time."""
    import torch.distributed as dist
    from det_sam2_amd.parallel import ShardedVideoProcessor
    from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
    B, K, W = a.objects, a.steps, a.warmup
    b = max(K // 2, 1)
    K = 2 * b
    warm_rounds = 1 + (max(W - b, 0) + K - 1) // K
    vp = ShardedVideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(B), skip_classes=set(), predictor=pred,
                               frame_buffer_size=b, detect_interval=b, max_frame_num_to_track=K, max_inference_state_frames=K)
    per_round = world * b
    frames = [synthetic_frame(t, 4242) for t in range((warm_rounds + 2) * per_round)]       # host uint8, same on every rank
    t = 0
    for _ in range(warm_rounds * per_round):
        vp.process_frame(t, frames[t])
        t += 1
    torch.cuda.synchronize()
    tracked0, enc0 = pred.stats["tracked_frames"], pred.stats["encoder_runs"]
    pred.trace = []
    pred.hip.profile_enable(True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(per_round):
        vp.process_frame(t, frames[t])
        t += 1
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    pred.hip.profile_enable(False)
    tracked = pred.stats["tracked_frames"] - tracked0
    cond_yield = K - tracked          # conditioning frames inside the window are yielded without tracking
    # one more round, UNTIMED, with a device synchronisation at every phase boundary: where a round's wall time goes
    # (ingest / encode / detect / prompt / gather = the all-gathers / ring_wait = what is left of the posted hand-off /
    # propagate), per rank - so that the first real multi-GPU run explains itself
    vp.profile_rounds = True
    for _ in range(per_round):
        vp.process_frame(t, frames[t])
        t += 1
    vp.profile_rounds = False
    split = vp.round_times[-1] if vp.round_times else {}
    names = ["ingest", "encode", "detect", "prompt", "gather", "ring_wait", "propagate"]
    sp = torch.tensor([split.get(k, 0.0) for k in names], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    sp_all = [torch.zeros_like(sp) for _ in range(world)]
    dist.all_gather(sp_all, sp)
    stats = torch.tensor([dt, tracked, pred.stats["encoder_runs"] - enc0], dtype=torch.float64,
                         device=dev if dist.get_backend() == "nccl" else "cpu")
    tmax = stats.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = stats.clone()
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    dt = float(tmax[0].item())
    ca_ms, ca_n = pred.hip.profile_read("kernel.cross_attention")
    stage_ms = {}
    for tag in ("stage.image_encoder", "stage.memory_attention", "stage.sam_heads", "stage.memory_encoder", "kernel.self_attention"):
        ms, n = pred.hip.profile_read(tag)
        stage_ms[tag] = round(ms / max(K, 1), 3)
    if rank == 0:
        nks = [tr["nk"] for tr in pred.trace]
        flops = sum(cross_attention_flops(B, nk) for nk in nks) * cfg.mem_attn_layers
        achieved = flops / (ca_ms * 1e-3) / 1e12 if ca_n else None
        comm = {}
        for _, op, nbytes in vp.comm_log[-8:]:
            comm[op] = nbytes
        total_tracked = float(tsum[1].item())
        out = {
            "metric": metric_name(cfg.name, B),
            "value": total_tracked / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.precision],
            "data": "synthetic",
            "config": {"workload": f"{cfg.name}, ONE stream of 1024x1024 frames, {B} objects, propagate passes sharded over {world} GPUs "
                                   f"(pass k -> rank k mod {world}); Det-SAM2 schedule scaled to buffer/detect {b}, track/keep {K}; one timed round "
                                   f"= {per_round} new stream frames from HOST uint8 (H2D + ingest on every rank), {K} frames visited per rank of which "
                                   f"{K - cond_yield} tracked and {cond_yield} conditioning; value counts tracked frames; bank of up to 3 conditioning + 6 "
                                   f"non-conditioning frames (Nk {min(nks) if nks else 0}..{max(nks) if nks else 0}); synthetic checkpoint seed 0",
                       "objects": B, "Nk_max": max(nks) if nks else 0, "frames_per_rank": K, "encode_batch": pred.encode_batch,
                       "parallelism": f"one stream, pass-sharded over {world} ranks; RCCL per round: 2 small fixed-size all_gathers (detections "
                                      f"fp64 [256,7], entry metadata int32 [32,2]), 1 all_gather of the new cond entries, 1 posted ring "
                                      f"send/recv of the feature pyramids (waited for before the propagation)",
                       "comm_bytes_per_round": comm,
                       "round_time_split_s_by_rank": [{k: round(float(v), 4) for k, v in zip(names, x.tolist())} for x in sp_all],
                       "encoder_runs_per_round_all_ranks": float(tsum[2].item()), "new_frames_per_round": per_round},
            "stream_fps": per_round / dt,
            "roofline": {"bound": "mfma", "kernel": f"memory cross-attention ({cross_kernel_name(a.precision)}), 1 launch/layer, per-frame Nk from the bank trace",
                         "achieved": achieved, "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                         "frac": None if achieved is None else achieved / PEAK_TFLOPS[a.precision], "traffic": None,
                         "avg_launch_ms": ca_ms / max(ca_n, 1), "launches": ca_n,
                         "note": "rank 0; achieved = sum over tracked frames of 4 * 2*B*4096*Nk*(256+64) / summed HIP-event time"},
            "ms_per_step_by_stage": stage_ms,
        }
        # SURVEY 8(d)'s path-level figure for the sharded stream: per tracked frame F_enc is amortised over the frames a rank ENCODES
        # (every stream frame once per stream), the tracking part is priced at each frame's own Nk
        if cfg.name in F_ENC_GFLOP and nks:
            enc_runs = float(tsum[2].item())
            pf_total = (F_ENC_GFLOP[cfg.name] * enc_runs + world * sum(B * (116.0 + 0.017039 * nk + 3.64 + 11.61) for nk in nks)) * 1e9
            out["roofline_path"] = {"bound": "mfma", "achieved": pf_total / dt / 1e12, "peak": PEAK_TFLOPS[a.precision] * world, "unit": "TFLOP/s",
                                    "frac": pf_total / dt / 1e12 / (PEAK_TFLOPS[a.precision] * world),
                                    "note": "F_enc * encoder runs of all ranks + sum over tracked frames of B * (F_ma(Nk) + 3.64 + 11.61) GFLOP (rank 0's "
                                            "Nk trace taken for every rank), over the round's max-over-ranks time; peak = n_gpus * the dense bf16 figure"}
        print(json.dumps(out))


def timed_tracking(pred, B, K, W, seed, dev, world, extra_frames=0, profile=True):
    """The timed region of the N = 1 / replica bench: steady-state bank, then EXACTLY K tracked frames (every one encoded inside
    the region), packed masks copied to pinned host memory.  -> (seconds, generator, state, bank size)."""
    import torch.distributed as dist
    from det_sam2_amd.parallel import allgather_cond_entries
    from det_sam2_amd.synth import synthetic_box, synthetic_frame
    n_frames = 1 + PREFILL + W + K + extra_frames
    frames = torch.from_numpy(np.stack([synthetic_frame(t, seed) for t in range(n_frames)])).to(dev)
    st = pred.init_state(frames)
    del frames
    for o in range(B):
        pred.add_new_points_or_box(st, 0, o, box=synthetic_box(o % 16, 0, seed))
    gen = pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=n_frames, reverse=False, output="packed")
    hv, wv = st["video_height"], st["video_width"]
    host = torch.empty((K, B, hv, (wv + 7) // 8), dtype=torch.uint8).pin_memory()
    next(gen)                                   # frame 0: conditioning frame (no tracking)
    for _ in range(PREFILL + W):                # fill the bank to steady state + warmup
        next(gen)
    torch.cuda.synchronize()
    # the feature cache batch-encodes upcoming frames: drop whatever the warm-up pre-encoded so that every one of
    # the K timed frames is encoded inside the timed region (exactly K encoder runs are asserted below)
    for t in [t for t in st["cached_features"] if t > PREFILL + W]:
        st["cached_features"].pop(t)
    for t in [t for t in (st.get("_pending_features") or {}) if t > PREFILL + W]:   # (DS2_ASYNC_ENCODE=1: encoded ahead)
        st["_pending_features"].pop(t)
    # ... and keep the batched encoder from running ahead into the GEMM-probe frames that follow the timed ones
    full_order = st["_encode_order"]
    st["_encode_order"] = [t for t in full_order if t <= PREFILL + W + K]
    enc0 = pred.stats["encoder_runs"]
    nk = 4096 * 7 + 4 * 16
    assert pred.trace is None
    pred.trace = []
    if profile:
        pred.hip.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world > 1:   # the one data-path exchange of the pass-sharded design: this pass' cond-frame bank entry
        allgather_cond_entries(st["output_dict"]["cond_frame_outputs"][0])
    for i in range(K):
        _, _, bits = next(gen)
        host[i].copy_(bits, non_blocking=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if profile:
        pred.hip.profile_enable(False)
    assert all(tr["nk"] == nk for tr in pred.trace), [tr["nk"] for tr in pred.trace]
    assert pred.stats["encoder_runs"] - enc0 == K, (pred.stats, enc0, K)   # every timed frame was encoded in the timed region
    st["_encode_order"] = full_order
    pred.trace = None
    return dt, gen, st, nk


def hole_filling_leg(cfg, sd, dev, B, K, W, precision):
    """The SHIPPING default (VERDICT r5 missing #3): the predictor as build_sam2_video_predictor builds it - build_sam.py:126-135
    appends fill_hole_area=8, and sam2_video_predictor.py:1343-1346 then runs fill_holes_in_mask_scores on every inferred frame
    (on a GPU; the CPU reference skips it, misc.py:389-391, which is why the headline and its goldens run without).  Same
    timed region as the headline, with ds2_fill_holes (connected components of the 256 x 256 background + the area test) on
    every tracked frame."""
    from det_sam2_amd.build_sam import build_sam2_video_predictor
    pred = build_sam2_video_predictor(cfg.name, {"model": sd}, device=dev, max_batch=B)
    assert pred.fill_hole_area == 8
    pred.hip.set_precision(precision)
    dt, gen, st, nk = timed_tracking(pred, B, K, W, 0, dev, 1, profile=False)
    del gen, st
    return {"value": K / dt, "unit": "frames/s", "ms_per_step": dt / K * 1e3, "fill_hole_area": 8, "steps": K,
            "built_by": "det_sam2_amd.build_sam.build_sam2_video_predictor (apply_postprocessing=True, the reference's default)",
            "parity": "tests/test_hip_measured_shape.py::test_hole_filling_default_at_measured_shape vs the oracle fixture "
                      "oracle_fill8_large_b16 (the CPU reference cannot fill: its CC kernel is CUDA-only)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-stream", action="store_true", help="skip the VideoProcessor.run stream-level measurement (stream_fps)")
    ap.add_argument("--stream-frames", type=int, default=150)
    ap.add_argument("--gemm-table", default=None, help="write the per-shape GEMM table (HIP events) to this file")
    ap.add_argument("--model", default="sam2.1_hiera_l")
    ap.add_argument("--objects", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hole-filling-leg", action="store_true",
                    help="skip the second timed leg with the shipping default fill_hole_area=8 (value_with_hole_filling)")
    ap.add_argument("--precision", default=os.environ.get("DS2_BENCH_PREC", "bf16x3k"), choices=["fp32", "bf16x3", "bf16x3k"])
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: N independent streams, one per GPU (BASELINE config 5) instead of ONE stream sharded by pass (config 4)")
    a = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DS2_BENCH_BACKEND=gloo + DS2_BENCH_ONE_DEVICE=1: dry run of the multi-rank code path on a single-GPU box (all
    # ranks on cuda:0, collectives staged through the host) - a functional check, not a measurement
    backend = os.environ.get("DS2_BENCH_BACKEND", "nccl")
    if os.environ.get("DS2_BENCH_ONE_DEVICE"):
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            else:
                dist.init_process_group(backend)
            probe = torch.ones(1, device=f"cuda:{local}")      # first collective: RCCL communicator set-up over xGMI
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert int(probe.item()) == world
        except Exception as e:     # the first real multi-GPU run must explain itself: one JSON line per failing rank
            print(json.dumps({"error": "distributed init failed", "rank": rank, "local_rank": local, "world_size": world,
                              "backend": backend, "exception": f"{type(e).__name__}: {e}",
                              "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                              "MASTER_ADDR": os.environ.get("MASTER_ADDR"), "visible_gpus": torch.cuda.device_count()}), flush=True)
            raise
    dev = f"cuda:{local}"
    if os.environ.get("DS2_BENCH_HIPRIO"):   # experiment: the tracking chain on a high-priority stream (DS2_ASYNC_ENCODE=1 puts the
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))   # encoder on a default-priority side stream)

    from det_sam2_amd.config import resolve_config
    from det_sam2_amd.parallel import allgather_cond_entries
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from det_sam2_amd.synth import synthetic_box, synthetic_frame
    from det_sam2_amd.weights import synthetic_state_dict

    cfg = resolve_config(a.model)
    B, K, W = a.objects, a.steps, a.warmup
    sd = synthetic_state_dict(cfg, 0)
    pred = SAM2VideoPredictor(cfg, sd, dev, max_batch=B)
    pred.hip.set_precision(a.precision)
    if world > 1 and not a.replicas:
        bench_sharded(a, pred, cfg, world, rank, dev)
        dist.destroy_process_group()
        return
    seed = 1000 * rank   # every rank (= its own pass shard) sees different frames
    dt, gen, st, nk = timed_tracking(pred, B, K, W, seed, dev, world, extra_frames=GEMM_PROBE)
    if world > 1:
        tmax = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ca_ms, ca_n = pred.hip.profile_read("kernel.cross_attention")
    stage_ms = {}
    for tag in ("stage.image_encoder", "stage.memory_attention", "stage.sam_heads", "stage.memory_encoder", "kernel.self_attention"):
        ms, n = pred.hip.profile_read(tag)
        stage_ms[tag] = round(ms / max(K, 1), 3)
    gemm, by_kernel = kernel_probe(pred, gen, st, PREFILL + W + K, a.gemm_table) if rank == 0 else (None, {})
    del gen, st
    stream = None
    if rank == 0 and world == 1 and not a.no_stream:
        stream = stream_fps(pred, B, a.stream_frames)
    filled = None
    encode_batch, async_encode = pred.encode_batch, bool(pred.async_encode)
    if rank == 0 and world == 1 and not a.no_hole_filling_leg:
        del pred
        torch.cuda.empty_cache()
        filled = hole_filling_leg(cfg, sd, dev, B, K, W, a.precision)
    if rank == 0:
        peak = PEAK_TFLOPS[a.precision]
        achieved = cross_attention_flops(B, nk) / (ca_ms / max(ca_n, 1) * 1e-3) / 1e12 if ca_n else None
        cross = {"bound": "mfma",
                 "kernel": f"memory cross-attention ({cross_kernel_name(a.precision)}), 1 launch/layer",
                 "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": None if achieved is None else achieved / peak,
                 "traffic": None, "traffic_from_committed_pmc": committed_pmc_traffic("cross_attention", B, nk, a.precision),
                 "algorithmic_bytes": cross_attention_bytes(B, nk, precision=a.precision),
                 "algorithmic_bytes_definition": "ONE definition for the launch: the one-kernel form - Q fp32 + K plane + V^T planes read once, "
                                                 "output planes written once; traffic_from_committed_pmc is k_attention_x4a's own (Q as fp16 "
                                                 "fragments + K plane + V^T planes in, unnormalised rows + (max, sum) out)",
                 "avg_launch_ms": ca_ms / max(ca_n, 1), "launches": ca_n,
                 "peak_sustained_measured": SUSTAINED_MFMA_TFLOPS,
                 "frac_of_sustained": None if achieved is None else achieved / SUSTAINED_MFMA_TFLOPS,
                 "sustained_note": "the fully loaded chip is POWER-limited: this kernel's own MFMA stream (real operands, nothing else) runs "
                                   "at 1.67 PFLOP/s on 256 CUs and at the nominal 2.5 on 128 (profiles/r04_mfma_power_calibration.txt); "
                                   "`peak` stays the guide's dense figure",
                 "note": "(mode bf16x3k, 16 objects: one 'launch' = k_attention_x4a alone - its query pass is the epilogue of k_qproj_x4a, its "
                         "normalisation / merge the prologue of k_vo_merge since round 5; with a key split (few objects) the parts are merged there too) "
                         "achieved = algorithmic FLOPs 2*B*4096*Nk*(256+64) per launch / mean HIP-event launch time INSIDE the timed "
                         "region (with async_encode the next encoder batch shares the CUs for ~60 % of it: reads ~5 % longer than the "
                         "kernel alone, which `by_kernel` below gives); executed MFMA FLOPs per algorithmic FLOP: bf16x3 3.0 "
                         "(frac <= 1/3), bf16x3k 1.0"}
        # the dominant kernel = the one with the largest total time per tracked frame (HIP events, kernels alone)
        dom = None
        cand = {k: v for k, v in by_kernel.items() if v["tflops"] is not None}
        if cand:
            name = max(cand, key=lambda k: cand[k]["ms_per_frame"])
            v = cand[name]
            is_gemm = name.startswith("k_gemm_split") or name.startswith("k_gemm_x4g")
            dom = {"bound": "mfma", "kernel": name, "achieved": v["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": v["tflops"] / peak,
                   "peak_sustained_measured": SUSTAINED_MFMA_TFLOPS, "frac_of_sustained": v["tflops"] / SUSTAINED_MFMA_TFLOPS,
                   "traffic": None,
                   "traffic_from_committed_pmc": committed_pmc_traffic(name.split()[0].split("<")[0] if is_gemm else "cross_attention", B, nk, a.precision),
                   "avg_launch_ms": v["avg_launch_ms"], "launches_per_frame": v["launches_per_frame"], "ms_per_frame": v["ms_per_frame"],
                   "algorithmic_flops_per_launch_mean": v["flops_per_launch"],
                   "note": "the kernel with the largest total time per tracked frame (per-kernel HIP-event brackets over one encoder "
                           f"batch of {GEMM_PROBE} frames right after the timed region, async encoder off = the kernel alone); achieved = "
                           "sum of algorithmic FLOPs of its launches / sum of their durations"
                           + ("; a bf16x3 GEMM executes 3 MFMA FLOPs per algorithmic FLOP, so frac <= 1/3" if is_gemm else "")}
        pf = path_flops(cfg.name, B, nk)
        out = {
            "metric": metric_name(cfg.name, B),
            "value": world * K / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.precision],
            "data": "synthetic",
            "config": {"workload": f"{cfg.name} propagate_in_video, {B} objects, 1024x1024 uniform-noise frames, "
                                   f"7-frame memory bank + 16 object pointers (Nk={nk}), synthetic checkpoint seed 0, "
                                   f"encoder run on every tracked frame, packed masks copied to host; frames are PRE-RESIDENT "
                                   f"in HBM as fp16 (H2D of 3 MiB/frame + ds2_ingest_frames are outside the timed region; the "
                                   f"stream_fps leg below starts from host uint8 frames)",
                       "objects": B, "Nk": nk, "frames_per_rank": K, "encode_batch": encode_batch,
                       "async_encode": async_encode,
                       "parallelism": "single GPU" if world == 1 else f"{world} independent replica streams (BASELINE config 5) + one RCCL all-gather of a cond entry"},
            "roofline": dom if dom is not None else cross,
            "roofline_cross_attention": cross,
            "roofline_path": None if pf is None else {
                "bound": "mfma", "achieved": pf * world * K / dt / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": pf * world * K / dt / 1e12 / peak,
                "flops_per_frame": pf,
                "note": "SURVEY 8(d): F(config) * tracked frames / wall time, F = F_enc + B*(F_ma(Nk) + 3.64 + 11.61) GFLOP as torch's "
                        "FlopCounterMode counts the reference (it materialises V: 2*(256+256) per score instead of the 2*(256+64) executed here)"},
            "by_kernel": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                          for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms_per_frame"])},
            "ms_per_step_by_stage": stage_ms,
        }
        if gemm is not None:
            out["roofline_gemm"] = gemm
        if filled is not None:
            out["value_with_hole_filling"] = filled["value"]
            out["hole_filling"] = filled
        if stream is not None:
            out["stream_fps"] = stream["stream_fps"]
            out["stream"] = stream
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, B)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
