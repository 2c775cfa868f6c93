#!/usr/bin/env python
"""The register-transposed epilogue of the key projection (k_gemm_split_k64t) against the slab epilogue (DS2_GEMM_K64T=0): the memory
attention at the bench size (16 objects, 7-frame bank + 16 pointers; also a ragged bank) must agree BIT FOR BIT; kernel times from the
library's HIP-event brackets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from det_sam2_amd.config import resolve_config
from det_sam2_amd.hip_model import HipSam2
from det_sam2_amd.weights import synthetic_state_dict

cfg = resolve_config("sam2.1_hiera_t")
hm = HipSam2(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=16)
d = hm.device
ok = True
for prec in ("bf16x3k", "bf16x3"):
    hm.set_precision(prec)
    for B, NF, NP in ((16, 7, 16), (16, 7, 13), (4, 2, 3)):
        g = torch.Generator().manual_seed(5 + NP)
        curr = torch.randn(4096, 256, generator=g).to(d)
        feats = [(torch.randn(B, 4096, 64, generator=g).to(torch.bfloat16).to(d), 6 - i) for i in range(NF)]
        ptrs = [(torch.randn(B, 256, generator=g).to(d), float(i) / 15.0) for i in range(NP)]
        mem_d, pos_d = hm.bank_assemble(B, feats, ptrs)
        outs, times = {}, {}
        for mode in ("0", "1"):
            os.environ["DS2_GEMM_K64T"] = mode
            hm.memory_attention(B, curr, mem_d, pos_d, 4 * NP)
            torch.cuda.synchronize()
            hm.profile_enable(True, gemm_shapes=True)
            for t in hm.profile_tags():
                hm.profile_read(t)
            outs[mode] = hm.memory_attention(B, curr, mem_d, pos_d, 4 * NP).clone()
            torch.cuda.synchronize()
            for t in hm.profile_tags():
                ms, n = hm.profile_read(t)
                if t.startswith("kern k_gemm_split_k64") and t.split()[3] == "256" and n:
                    t0, n0 = times.get(mode, (0.0, 0))
                    times[mode] = (t0 + ms * 1e3, n0 + n)
            hm.profile_enable(False)
        same = torch.equal(outs["0"], outs["1"])
        ok &= same
        tt = {m: (round(v[0] / max(v[1], 1), 1) if isinstance(v, tuple) else v) for m, v in times.items()}
        print(f"{prec} B={B} Nk={4096 * NF + 4 * NP}: {'BIT-IDENTICAL' if same else 'DIFFERENT max|d| %g' % float((outs['0'] - outs['1']).abs().max())}  k64 kernels us/launch {tt}", flush=True)
print("K64T CHECK", "PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
