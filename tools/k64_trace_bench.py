#!/usr/bin/env python
"""Timeline of the memory attention's key projection (k_gemm_split_k64 with RoPE, hi plane out) inside bench.py's workload,
from a -DDS2_K64_TRACE=2 build:  DS2_LIB=.../ab_k64trace2.so python tools/k64_trace_bench.py"""
import ctypes, os, subprocess, sys
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
buf = np.zeros((8, 256), dtype=np.uint64)
assert lib.ds2_debug_k64_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
per = 2 + 2 * 4
for w in (0, 1, 4):
    n = int((t[w] > 0).sum()) // per
    x = t[w, :n * per].reshape(n, per)
    print(f"wave {w}: tiles {n}")
    for i in range(n):
        d = np.diff(x[i])
        print(f"  tile {i}: mfma {d[0]:6d} | " + " ".join(f"park {d[1+2*k]:5d} stream {d[2+2*k]:6d}" for k in range(4)))
