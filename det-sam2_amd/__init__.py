"""det-sam2_amd: MI355X-native (gfx950) SAM 2.1 video-predictor hot path of Det-SAM2.

Public surface mirrors the reference (``sam2.build_sam.build_sam2_video_predictor``,
``SAM2VideoPredictor``, ``det_sam2_RT.VideoProcessor``); the arithmetic is hand-written HIP
behind the C-ABI declared in ``include/detsam2_hip.h``.
"""
__all__ = ["config", "weights"]
