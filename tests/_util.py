"""Shared helpers for the GPU parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(name, **kw):
    """Append a metrics line to gpurun_out/metrics.jsonl (scratch; merged back by gpurun)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "metrics.jsonl"), "a") as f:
            row = {k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in kw.items()}
            if os.environ.get("DS2_VARIANT"):
                row["variant"] = os.environ["DS2_VARIANT"]
            f.write(json.dumps({"test": name, **row}) + "\n")
    except Exception:
        pass


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))
