"""The assembly bodies of the persistent GEMM (det-sam2_amd/csrc/gemm_x4g_body_<cfg>_<epi>.inc) are GENERATED: the committed files must
be what tools/gen/gen_gemm_x4g.py writes today (no hand edits, no stale schedule), and the generator's hazard lint must pass."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", ["42", "23"])
@pytest.mark.parametrize("epi", ["e1", "e2", "e3"])
def test_committed_body_is_the_generators_output(tmp_path, cfg, epi):
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4G_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_gemm_x4g.py"), str(out), cfg, epi], check=True, env=env,
                   capture_output=True)
    committed = open(os.path.join(ROOT, "det-sam2_amd", "csrc", f"gemm_x4g_body_{cfg}_{epi}.inc")).read()
    assert out.read_text() == committed
