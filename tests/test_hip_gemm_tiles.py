"""Forced GEMM tile kernels (DS2_GEMM_TILE: 1 = 128x128, 3 = 256x128 three-stage ring, 5 = 256x256 LDS-DMA, 10 = persistent
256x256) on shapes the heuristic would not hand them: every kernel of the family must be correct on every shape it accepts, not
only on the ones the default dispatch picks (forced-tile paths were where wrong-address bugs hid).  The
switch is read once per process, so each tile runs in a subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from det_sam2_amd.hip_model import HipOps
o = HipOps("cuda:0")
o.set_precision("bf16x3")
worst = 0.0
# (M, N, K, act, residual, r_mod): ragged M / N / K, residual and broadcast residual, GELU, more tiles than CUs
for (M, N, K, act, use_r, r_mod) in [(300, 200, 96, 0, False, 0), (1000, 576, 144, 2, True, 0), (4100, 256, 64, 0, True, 4096),
                                     (70000, 320, 72, 1, False, 0), (513, 1152, 1152, 0, True, 0)]:
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    R = torch.randn(r_mod if r_mod else M, N, generator=g) if use_r else None
    ref = A.double() @ W.double().T + b.double()
    ref = [lambda x: x, torch.relu, torch.nn.functional.gelu][act](ref)
    if use_r:
        ref = ref + (R.double()[torch.arange(M) %% r_mod] if r_mod else R.double())
    d = o.device
    got = o.op_gemm(A.to(d), W.to(d), b.to(d), act, None, None if R is None else R.to(d), r_mod)
    torch.cuda.synchronize()
    e = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    assert e < 2e-5, (M, N, K, act, e)
print("worst", worst)
""" % ROOT


@pytest.mark.parametrize("tile", [1, 3, 5, 10])
def test_forced_gemm_tile(tile):
    env = dict(os.environ, DS2_GEMM_TILE=str(tile))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst" in r.stdout
