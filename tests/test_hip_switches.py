"""Every runtime switch of the library at its NON-default value (README "Runtime switches"; VERDICT r3 weak #11: a switch is
either tested or deleted): config 1 and the 16-object scenario against their reference goldens, one subprocess per switch (the
switches are read once per process).  The defaults are what every other GPU test runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SWITCHES = ["DS2_ASYNC_ENCODE=0", "DS2_ATTN_HG=0", "DS2_ATTN_KSPLIT=0", "DS2_ATTN_NO_VLO_SKIP=1",
            "DS2_GEMM_K64=0", "DS2_GEMM_PP256=0", "DS2_GEMM_SKINNY=0", "DS2_GEMM_X4G=0", "DS2_GEMM_K64T=0", "DS2_MA_FOLD_VO=0",
            "DS2_ME_COL_PLANES=0", "DS2_ME_FUSE_UP=0", "DS2_MLP_FUSED=0", "DS2_ENCODE_BATCH=3", "DS2_ATTN_X4A=0", "DS2_F16X2=0", "DS2_MA_FUSE_LN=0", "DS2_ATTN_WINLDS=0", "DS2_MLP_HSPLIT=0", "DS2_BANK_DIRECT=0", "DS2_MA_LN3_FUSE=0", "DS2_ME_LN_FUSE=0", "DS2_MA_QFUSE=0", "DS2_MA_QKVFUSE=0", "DS2_MA_VOFUSE=0"]


@pytest.mark.parametrize("switch", SWITCHES)
def test_non_default_switch_keeps_parity(switch):
    k, _, v = switch.partition("=")
    env = dict(os.environ)
    env[k] = v
    r = subprocess.run([sys.executable, os.path.join(HERE, "_switch_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (switch, r.stdout[-1500:], r.stderr[-1500:])
    worst = float([l for l in r.stdout.splitlines() if l.startswith("WORST")][-1].split()[1])
    assert worst <= 1e-3, (switch, worst)
