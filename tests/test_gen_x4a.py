"""The assembly body of the cross-attention kernel (det-sam2_amd/csrc/attention_x4a_body.inc) is GENERATED: the committed file
must be what tools/gen/gen_attention_x4a.py writes today (no hand edits, no stale schedule)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_body_is_the_generators_output(tmp_path):
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4A_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_attention_x4a.py"), str(out)], check=True, env=env,
                   capture_output=True)
    committed = open(os.path.join(ROOT, "det-sam2_amd", "csrc", "attention_x4a_body.inc")).read()
    assert out.read_text() == committed
