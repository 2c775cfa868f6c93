"""Harness that imports the *reference* (read-only, /root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_goldens.py`` and by the
``ref``-marked pinning tests, which are skipped whenever ``/root/reference`` is
absent (i.e. always on the GPU box).  Nothing in the product imports this.

The reference needs a few third-party modules that are not installed here
(hydra, omegaconf, cv2, ultralytics, IPython, pympler).  We do not edit the
reference; we put harness-level stand-ins into ``sys.modules`` before importing
it (recipe: SURVEY.md section 8c):

* ``hydra``/``omegaconf``: only touched at import time (``sam2/__init__.py:7-11``,
  ``sam2/build_sam.py:11-13``).  The model is instead instantiated by walking the
  YAML ``_target_`` tree with PyYAML (``instantiate_from_yaml`` below), applying the
  same five overrides ``build_sam2_video_predictor`` appends
  (``sam2/build_sam.py:121-135``).
* ``cv2``: ``resize`` is only ever called with a 1024x1024 source in our runs, where
  it is the identity; ``cvtColor`` is a channel flip.
* ``ultralytics.YOLO``: replaced by a scripted detector that replays boxes we give it
  (output contract from ``det_sam2_RT.py:228-238``).
* ``readerwriterlock`` (``Det_SAM2_pipeline.py:10,71``): ``rwlock.RWLockWrite().gen_wlock()`` as a plain mutex - the
  pipeline only ever takes the write lock.
* ``cv2.VideoCapture`` (``Det_SAM2_pipeline.py:115-131``): replays the frames registered under a source name
  (``ScriptedCapture.sources``), as BGR - the pipeline converts every frame back with ``cvtColor(BGR2RGB)``.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch
import yaml

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sam2"))


class ScriptedDetector:
    """Stand-in for ``ultralytics.YOLO``: replays a per-call script of boxes.

    ``script`` is a list (one entry per *selected frame*, in call order) of lists of
    ``(xyxy, cls, conf)``.
    """

    script: list = []
    cursor: int = 0

    def __init__(self, *_a, **_k):
        pass

    def __call__(self, frames, stream=True, conf=0.0, iou=0.0, verbose=False):
        for _ in frames:
            dets = ScriptedDetector.script[ScriptedDetector.cursor]
            ScriptedDetector.cursor += 1
            boxes = []
            for xyxy, cls, cf in dets:
                b = types.SimpleNamespace(
                    xyxy=torch.tensor([xyxy], dtype=torch.float32),
                    cls=torch.tensor([float(cls)]),
                    conf=torch.tensor([float(cf)]),
                )
                boxes.append(b)
            yield types.SimpleNamespace(boxes=boxes)


class ScriptedCapture:
    """Stand-in for ``cv2.VideoCapture``: ``sources[name]`` is a list of RGB uint8 frames; ``read()`` hands them out
    as BGR and reports the end of the stream with ``(False, None)`` like OpenCV does."""

    sources: dict = {}

    def __init__(self, source):
        self.frames = ScriptedCapture.sources.get(source)
        self.pos = 0

    def isOpened(self):
        return self.frames is not None

    def read(self):
        if self.pos >= len(self.frames):
            return False, None
        f = self.frames[self.pos]
        self.pos += 1
        return True, f[..., ::-1].copy()

    def release(self):
        pass


class _WriteLock:
    def __init__(self):
        import threading
        self._m = threading.Lock()

    def gen_wlock(self):
        return self._m

    def gen_rlock(self):
        return self._m


def install_shims() -> None:
    if "sam2" in sys.modules:
        return
    # --- hydra / omegaconf (import-time only)
    hydra = types.ModuleType("hydra")
    hydra.initialize_config_module = lambda *a, **k: None
    hydra.compose = lambda *a, **k: None
    hcore = types.ModuleType("hydra.core")
    hgh = types.ModuleType("hydra.core.global_hydra")

    class _GH:
        @staticmethod
        def instance():
            return types.SimpleNamespace(is_initialized=lambda: True)

    hgh.GlobalHydra = _GH
    hutils = types.ModuleType("hydra.utils")
    hutils.instantiate = lambda *a, **k: None
    omegaconf = types.ModuleType("omegaconf")
    omegaconf.OmegaConf = types.SimpleNamespace(resolve=lambda c: None)
    for name, mod in [("hydra", hydra), ("hydra.core", hcore), ("hydra.core.global_hydra", hgh),
                      ("hydra.utils", hutils), ("omegaconf", omegaconf)]:
        sys.modules[name] = mod
    # --- cv2
    cv2 = types.ModuleType("cv2")

    def _resize(img, size):
        assert img.shape[0] == size[1] and img.shape[1] == size[0], "harness cv2.resize: identity only"
        return img

    cv2.resize = _resize
    cv2.COLOR_RGB2BGR = 0
    cv2.COLOR_BGR2RGB = 1
    cv2.cvtColor = lambda img, code: img[..., ::-1].copy()
    cv2.VideoCapture = ScriptedCapture
    sys.modules["cv2"] = cv2
    # --- readerwriterlock (the pipeline's hand-off lock)
    rwl = types.ModuleType("readerwriterlock")
    rwl.rwlock = types.ModuleType("readerwriterlock.rwlock")
    rwl.rwlock.RWLockWrite = _WriteLock
    sys.modules["readerwriterlock"] = rwl
    sys.modules["readerwriterlock.rwlock"] = rwl.rwlock
    # --- ultralytics / IPython / pympler / frames2video
    ul = types.ModuleType("ultralytics")
    ul.checks = lambda: None
    ul.YOLO = ScriptedDetector
    sys.modules["ultralytics"] = ul
    ipy = types.ModuleType("IPython")
    disp = types.ModuleType("IPython.display")
    disp.clear_output = lambda *a, **k: None
    disp.display = lambda *a, **k: None
    disp.Image = object
    ipy.display = disp
    sys.modules["IPython"] = ipy
    sys.modules["IPython.display"] = disp
    pym = types.ModuleType("pympler")
    pym.asizeof = types.SimpleNamespace(asizeof=lambda *a, **k: 0)
    sys.modules["pympler"] = pym
    f2v = types.ModuleType("frames2video")
    f2v.frames_to_video = lambda *a, **k: None
    sys.modules["frames2video"] = f2v
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
        sys.path.insert(0, os.path.join(REFERENCE_ROOT, "det_sam2_inference"))
    import matplotlib

    matplotlib.use("Agg")


def _coerce(v):
    if isinstance(v, str):
        try:
            return float(v)  # PyYAML reads "1e-6" as a string; OmegaConf reads a float
        except ValueError:
            return v
    return v


def _instantiate(node):
    if isinstance(node, dict):
        if "_target_" in node:
            modname, clsname = node["_target_"].rsplit(".", 1)
            cls = getattr(importlib.import_module(modname), clsname)
            kwargs = {k: _instantiate(v) for k, v in node.items() if k != "_target_"}
            return cls(**kwargs)
        return {k: _instantiate(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_instantiate(v) for v in node]
    return _coerce(node)


def instantiate_from_yaml(config_file: str, state_dict=None, apply_postprocessing: bool = True):
    """Equivalent of ``build_sam2_video_predictor(config_file, device='cpu', apply_postprocessing=...)``
    (``sam2/build_sam.py:111-146``) without hydra."""
    install_shims()
    path = os.path.join(REFERENCE_ROOT, "sam2", config_file)
    with open(path) as f:
        cfg = yaml.safe_load(f)["model"]
    cfg["_target_"] = "sam2.sam2_video_predictor.SAM2VideoPredictor"
    if apply_postprocessing:      # the five overrides of build_sam.py:126-135
        cfg["sam_mask_decoder_extra_args"] = {
            "dynamic_multimask_via_stability": True,
            "dynamic_multimask_stability_delta": 0.05,
            "dynamic_multimask_stability_thresh": 0.98,
        }
        cfg["binarize_mask_from_pts_for_mem_enc"] = True
        cfg["fill_hole_area"] = 8
    torch.manual_seed(0)
    model = _instantiate(cfg)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict)  # strict, as build_sam.py:166-177
        assert not missing and not unexpected, (missing, unexpected)
    return model.to("cpu").eval()


def make_reference_video_processor(config_file: str, state_dict, **vp_kwargs):
    """Instantiate the reference ``VideoProcessor`` (det_sam2_RT.py:25) on CPU."""
    install_shims()
    import det_sam2_RT  # noqa: F401  (reference module, found through sys.path)

    det_sam2_RT.build_sam2_video_predictor = lambda cfg, ckpt: instantiate_from_yaml(cfg, state_dict)
    vp = det_sam2_RT.VideoProcessor(
        output_dir=vp_kwargs.pop("output_dir", "/tmp/ref_vp_out"),
        sam2_checkpoint=None,
        model_cfg=config_file,
        detect_model_weights=None,
        **vp_kwargs,
    )
    return vp
