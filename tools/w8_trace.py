#!/usr/bin/env python
"""Per-tile timeline of the memory cross-attention kernel (k_attention_w8<64,2,*>) inside bench.py's workload from a -DDS2_ATT_TRACE=1 (global attention)
or =2 (256-key windows) build:  DS2_LIB=.../ab_atttrace1.so python tools/w8_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-stream"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from det_sam2_amd import _capi
lib = ctypes.CDLL(_capi.LIB_PATH)
buf = np.zeros((2, 512), dtype=np.uint64)
assert lib.ds2_debug_w8_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
names = ["loads", "QK mfma", "softmax", "PV mfma", "stage", "barrier->next"]
for w in range(2):
    n = int((t[w] > 0).sum()) // 6
    x = t[w, :n * 6].reshape(n, 6)
    d = np.diff(np.concatenate([x.reshape(-1), x[-1:, -1]]))[: n * 6].reshape(n, 6)
    d[-1, -1] = 0
    print(f"wave {w * 4}: {n} tiles; mean cycles per segment over tiles 2..{n - 2}:")
    m = d[2:n - 1].mean(axis=0)
    print("   " + "  ".join(f"{names[i]} {m[i]:.0f}" for i in range(6)) + f"   | tile {m.sum():.0f}")
