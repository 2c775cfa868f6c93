"""CPU stand-in for HipSam2's stage interface (TEST INFRASTRUCTURE, never shipped): tiny deterministic torch arithmetic
with the real data FLOW - every output depends on every input the real stage reads - so that the host logic above the
C-ABI (predictor state machine, streaming driver, pass-sharded rounds and their collectives) can be exercised on CPU and
under gloo.  Shapes are shrunk (16 tokens, 8-d memory, 16x16 masks); nothing here approximates SAM 2 numerics."""
import numpy as np
import torch

TOK, MEM, PTR, SIDE, FEAT = 16, 8, 16, 256, 12     # the predictor hard-codes 256x256 low-res masks


class FakeHip:
    def __init__(self):
        self.device = torch.device("cpu")
        self.no_obj_ptr = torch.full((1, PTR), 0.25)
        self.calls = {"image_encoder": 0, "memory_attention": 0}

    def set_precision(self, mode):
        pass

    # ---- A3
    def ingest(self, u8):
        return (u8.permute(0, 3, 1, 2).to(torch.float32) / 255.0).to(torch.float16).contiguous()

    # ---- A4/A5
    def _enc(self, f):
        x = f.float()
        g = torch.nn.functional.adaptive_avg_pool2d(x, (4, 4)).reshape(3, TOK).T            # [16, 3]
        f2 = torch.cat([g, g * g, torch.sin(7 * g), torch.cos(3 * g)], 1)                    # [16, 12]
        return (f2[:4, :4].clone(), f2[:8, :6].clone(), f2.contiguous())

    def image_encoder(self, frame_f16):
        self.calls["image_encoder"] += 1
        return self._enc(frame_f16)

    def image_encoder_batch(self, frames_f16):
        self.calls["image_encoder"] += frames_f16.shape[0]
        return [self._enc(f) for f in frames_f16]

    # ---- A11
    def bank_assemble(self, B, mem_entries, ptr_entries):
        mems = [f.float() for f, _ in mem_entries] + [p.reshape(B, -1, MEM)[:, :1] for p, _ in ptr_entries]
        poss = [torch.full((B, TOK, MEM), 0.1 * (r + 1)) for _, r in mem_entries] + \
               [torch.full((B, 1, MEM), float(q)) for _, q in ptr_entries]
        assert all(f.shape == (B, TOK, MEM) for f, _ in mem_entries)
        return torch.cat(mems, 1), torch.cat(poss, 1)

    # ---- A12 (order-sensitive in the keys, like RoPE'd attention)
    def memory_attention(self, B, curr, memory, memory_pos, n_ptr_tok):
        self.calls["memory_attention"] += 1
        nk = memory.shape[1]
        w = torch.linspace(0.5, 1.5, nk).reshape(1, nk, 1)
        ctx = ((memory + memory_pos) * w).mean(1)                                            # [B, MEM]
        return curr[None] + torch.tanh(ctx).repeat(1, 2)[:, None, :FEAT] * 0.5 + 0.01 * n_ptr_tok

    # ---- A7/A8
    def sam_heads(self, B, pix, f0, f1, coords=None, labels=None, multimask=False, pix_bcast=False, add_no_mem_embed=False,
                  mask_inputs=None):
        if pix_bcast:
            pix = pix[None].expand(B, -1, -1)
        s = pix.sum((1, 2)) + f0.sum() * 0.1 + f1.sum() * 0.01 + (0.3 if add_no_mem_embed else 0.0) + (0.2 if multimask else 0.0)
        if coords is not None and coords.numel():
            s = s + coords.reshape(B, -1).sum(1) * 1e-3
        if mask_inputs is not None:
            s = s + mask_inputs.mean((1, 2))
        yy, xx = torch.meshgrid(torch.arange(SIDE), torch.arange(SIDE), indexing="ij")
        low = torch.sin(s.reshape(B, 1, 1) + 0.05 * yy + 0.03 * xx) * 4.0
        ptr = torch.tanh(pix.mean(1)).repeat(1, 2)[:, :PTR] + 0.1 * s.reshape(B, 1)
        obj = torch.ones(B)
        return low.contiguous(), ptr.contiguous(), obj, torch.ones(B)

    # ---- F3
    def use_mask_as_output(self, B, f2, f0, f1, mask):
        low = torch.nn.functional.adaptive_avg_pool2d(mask[:, None] * 20.0 - 10.0, (SIDE, SIDE))[:, 0]
        ptr = torch.tanh(f2.mean(0)).repeat(2)[None, :PTR].expand(B, -1) * mask.mean((1, 2)).reshape(B, 1)
        obj = torch.where(mask.flatten(1).any(1), 10.0, -10.0)
        return low.contiguous(), ptr.contiguous(), obj

    # ---- A13
    def memory_encoder(self, B, f2, low, obj, binarize):
        m = (low > 0).float() if binarize else torch.sigmoid(low)
        pooled = torch.nn.functional.adaptive_avg_pool2d(m[:, None], (4, 4)).reshape(B, TOK, 1)
        return (f2[None, :, :MEM] * 0.5 + pooled).to(torch.bfloat16).contiguous()

    # ---- A15
    def mask_output(self, low, hv, wv, want_logits=True, want_packed=True):
        lg = torch.nn.functional.interpolate(low[:, None], size=(hv, wv), mode="bilinear", align_corners=False)
        packed = torch.from_numpy(np.packbits((lg[:, 0] > 0).numpy(), axis=-1)) if want_packed else None
        return (lg if want_logits else None), packed


def fake_predictor():
    """SAM2VideoPredictor (the real state machine) over FakeHip."""
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor

    class P(SAM2VideoPredictor):
        def feature_shapes(self):
            return [(4, 4), (8, 6), (TOK, FEAT)]

        def entry_dims(self):
            return dict(tokens=TOK, mem_dim=MEM, ptr_dim=PTR, mask_side=SIDE, feat_dim=FEAT)

    p = P("sam2.1_hiera_t", None, device="cpu", hip=FakeHip())
    p.device = torch.device("cpu")
    p.hidden_dim = PTR          # width of the object pointers the fake heads emit
    return p
