"""Model hyper-parameters of the four SAM 2.1 variants on the Det-SAM2 hot path.

The reference composes these from Hydra YAML (``sam2/configs/sam2.1/sam2.1_hiera_{t,s,b+,l}.yaml``)
and ``build_sam2_video_predictor`` appends five overrides (``sam2/build_sam.py:121-135``).
Here they are plain data; ``resolve_config`` accepts the reference's config-file names so
``VideoProcessor(model_cfg='configs/sam2.1/sam2.1_hiera_l.yaml', ...)`` stays a drop-in call.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class HieraCfg:
    embed_dim: int
    num_heads: int
    stages: Tuple[int, ...]
    global_att_blocks: Tuple[int, ...]
    window_pos_embed_bkg_spatial_size: Tuple[int, int]
    window_spec: Tuple[int, ...]
    q_pool: int = 3          # hieradet.py:180
    q_stride: int = 2        # hieradet.py:181
    dim_mul: float = 2.0
    head_mul: float = 2.0

    def blocks(self):
        """Per-block geometry, following the constructor loop at hieradet.py:236-267.

        Returns a list of dicts: dim, dim_out, heads, window (0 = global), q_stride (0/2).
        """
        depth = sum(self.stages)
        stage_ends = [sum(self.stages[:i]) - 1 for i in range(1, len(self.stages) + 1)]
        q_pool_blocks = [x + 1 for x in stage_ends[:-1]][: self.q_pool]
        out = []
        dim, heads, cur_stage = self.embed_dim, self.num_heads, 1
        for i in range(depth):
            dim_out = dim
            window = self.window_spec[cur_stage - 1]
            if i in self.global_att_blocks:
                window = 0
            if i - 1 in stage_ends:
                dim_out = int(dim * self.dim_mul)
                heads = int(heads * self.head_mul)
                cur_stage += 1
            out.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window,
                            q_stride=self.q_stride if i in q_pool_blocks else 0))
            dim = dim_out
        return out

    @property
    def stage_ends(self):
        return [sum(self.stages[:i]) - 1 for i in range(1, len(self.stages) + 1)]

    @property
    def channel_list(self):
        b = self.blocks()
        return [b[i]["dim_out"] for i in self.stage_ends[::-1]]


@dataclass(frozen=True)
class ModelCfg:
    name: str
    trunk: HieraCfg
    image_size: int = 1024
    backbone_stride: int = 16
    d_model: int = 256            # FpnNeck.d_model / hidden_dim
    mem_dim: int = 64             # MemoryEncoder.out_dim
    num_maskmem: int = 7
    scalp: int = 1
    fpn_top_down_levels: Tuple[int, ...] = (2, 3)
    mem_attn_layers: int = 4
    mem_attn_ffn: int = 2048
    rope_theta: float = 10000.0
    max_obj_ptrs_in_encoder: int = 16         # sam2_base.py:49
    max_cond_frames_in_attn: int = 20         # sam2_base.py:42 (Det-SAM2 default)
    sigmoid_scale_for_mem_enc: float = 20.0
    sigmoid_bias_for_mem_enc: float = -10.0
    multimask_min_pt_num: int = 0
    multimask_max_pt_num: int = 1
    # overrides appended by build_sam2_video_predictor (build_sam.py:126-135)
    dynamic_multimask_via_stability: bool = True      # build_sam.py:128 (apply_postprocessing); MaskDecoder's own default is False
    dynamic_multimask_stability_delta: float = 0.05
    dynamic_multimask_stability_thresh: float = 0.98
    binarize_mask_from_pts_for_mem_enc: bool = True
    fill_hole_area: int = 8

    @property
    def feat_hw(self) -> int:
        return self.image_size // self.backbone_stride


_T = HieraCfg(96, 1, (1, 2, 7, 2), (5, 7, 9), (7, 7), (8, 4, 14, 7))
_S = HieraCfg(96, 1, (1, 2, 11, 2), (7, 10, 13), (7, 7), (8, 4, 14, 7))
_B = HieraCfg(112, 2, (2, 3, 16, 3), (12, 16, 20), (14, 14), (8, 4, 14, 7))
_L = HieraCfg(144, 2, (2, 6, 36, 4), (23, 33, 43), (7, 7), (8, 4, 16, 8))

CONFIGS = {
    "sam2.1_hiera_t": ModelCfg("sam2.1_hiera_t", _T),
    "sam2.1_hiera_s": ModelCfg("sam2.1_hiera_s", _S),
    "sam2.1_hiera_b+": ModelCfg("sam2.1_hiera_b+", _B),
    "sam2.1_hiera_l": ModelCfg("sam2.1_hiera_l", _L),
}
_ALIASES = {
    "sam2.1_hiera_tiny": "sam2.1_hiera_t",
    "sam2.1_hiera_small": "sam2.1_hiera_s",
    "sam2.1_hiera_base_plus": "sam2.1_hiera_b+",
    "sam2.1_hiera_large": "sam2.1_hiera_l",
}


def resolve_config(name_or_path) -> ModelCfg:
    """Accepts 'sam2.1_hiera_l', 'sam2.1_hiera_large', or the reference's YAML path
    'configs/sam2.1/sam2.1_hiera_l.yaml' (det_sam2_RT.py:667)."""
    if isinstance(name_or_path, ModelCfg):
        return name_or_path
    base = str(name_or_path).replace("\\", "/").split("/")[-1]
    if base.endswith(".yaml"):
        base = base[: -len(".yaml")]
    base = _ALIASES.get(base, base)
    if base not in CONFIGS:
        raise ValueError(f"unknown SAM 2.1 config {name_or_path!r}; known: {sorted(CONFIGS)}")
    return CONFIGS[base]
