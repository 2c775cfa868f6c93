"""The assembly bodies of the persistent GEMM (det-sam2_amd/csrc/gemm_x4g_body_<cfg>_<epi>.inc) are GENERATED: the committed files must
be what tools/gen/gen_gemm_x4g.py writes today (no hand edits, no stale schedule), and the generator's hazard lint must pass."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", ["42", "23", "23m"])      # 23m: the MX form of the 128 x 192 tile (round 6)
@pytest.mark.parametrize("epi", ["e1", "e2", "e3"])
def test_committed_body_is_the_generators_output(tmp_path, cfg, epi):
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4G_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_gemm_x4g.py"), str(out), cfg, epi], check=True, env=env,
                   capture_output=True)
    committed = open(os.path.join(ROOT, "det-sam2_amd", "csrc", f"gemm_x4g_body_{cfg}_{epi}.inc")).read()
    assert out.read_text() == committed
    if cfg == "23m":     # per K tile and wave: 6 blocks x (4 fp16 sub-steps + 2 scaled fp8 MFMAs); 8 drain bodies + plain + last
        assert committed.count("v_mfma_f32_32x32x16_f16") == 240 and committed.count("v_mfma_scale_f32_32x32x64_f8f6f4") == 120
        assert ("v_cvt_scalef32_pk_fp8_f32" in committed) == (epi == "e2")        # only the GELU form writes MX planes
        assert ("hwreg(HW_REG_MODE, 23, 1), 1" in committed) == (epi == "e2")     # ... with saturating conversions


def test_mx_scale_constants_agree():
    """the generator's static scales (tools/gen/gen_gemm_x4g.py MX_*) are those of the producers (csrc/common.h DS2_MX_*)"""
    import re
    gen = open(os.path.join(ROOT, "tools", "gen", "gen_gemm_x4g.py")).read()
    m = re.search(r"MX_EA, MX_LA, MX_EW, MX_LW = (-?\d+), (-?\d+), (-?\d+), (-?\d+)", gen)
    hdr = open(os.path.join(ROOT, "det-sam2_amd", "csrc", "common.h")).read()
    want = tuple(int(re.search(rf"#define DS2_MX_{k} \(?(-?\d+)\)?", hdr).group(1)) for k in ("EA", "LA", "EW", "LW"))
    assert tuple(int(x) for x in m.groups()) == want and want[0] + want[1] == want[2] + want[3]


def test_committed_gelu_mlp_loop_is_the_generators_output(tmp_path):
    """mlp256_x4m_gelu_body.inc (the loop with ds2_gelu's instruction sequence as the activation: memory encoder CXBlock) = X4M_ACT=gelu"""
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4M_")}
    env["X4M_ACT"] = "gelu"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_mlp256_x4m.py"), str(out)], check=True, env=env, capture_output=True)
    body = out.read_text()
    assert body == open(os.path.join(ROOT, "det-sam2_amd", "csrc", "mlp256_x4m_gelu_body.inc")).read()
    assert body.count("v_mfma_f32_32x32x16_f16") == 128 and body.count("v_exp_f32") == 32 and body.count("v_rcp_f32") == 32


def test_committed_mlp_loop_is_the_generators_output(tmp_path):
    """det-sam2_amd/csrc/mlp256_x4m_body.inc (the hidden loop of the fused MLP's ReLU / two-fp16-term form) = tools/gen/gen_mlp256_x4m.py
    today; the generator simulates the in-order LDS queue for every counted wait and checks that the queue at the end of the loop body
    equals the one at its start."""
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4M_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_mlp256_x4m.py"), str(out)], check=True, env=env, capture_output=True)
    assert out.read_text() == open(os.path.join(ROOT, "det-sam2_amd", "csrc", "mlp256_x4m_body.inc")).read()
    body = out.read_text()
    assert body.count("v_mfma_f32_32x32x16_f16") == 128 and body.count("s_barrier") == 9      # 64 units x 2; 8 steps + the prologue


@pytest.mark.parametrize("env_extra", [{"X4M_D": "4"}, {"X4M_D": "7"}, {"X4M_ILV": "1", "X4M_D": "4"}, {"X4M_WAIT1": "0", "X4M_NONOP": "0", "X4M_MED3": "0"}])
def test_mlp_loop_generator_variants_are_consistent(tmp_path, env_extra):
    """the schedule knobs (read-ahead depth, interleaved unit pairs, the first version's instruction mix) all pass the generator's own
    checks: queue state cyclic, hazard lint, 128 MFMAs"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4M_")}
    env.update(env_extra)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_mlp256_x4m.py"), str(tmp_path / "v.inc")], check=True, env=env,
                   capture_output=True)


@pytest.mark.parametrize("epi", ["e1", "e2", "e3"])
def test_mx_split_form_generates(tmp_path, epi):
    """the measured alternative of the MX body (X4G_MX_SPLIT=1: two barriers per K tile, the stage's two operand planes re-filled
    separately - tools/gen/gen_gemm_x4g.py body_mx2, profiles/r06_ab_mx_split.txt) passes the generator's own checks: every read of a
    stage issued before its barrier, hazard lint, the same MFMA mix"""
    out = tmp_path / "body.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("X4G_")}
    env["X4G_MX_SPLIT"] = "1"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_gemm_x4g.py"), str(out), "23m", epi], check=True, env=env, capture_output=True)
    body = out.read_text()
    assert body.count("v_mfma_f32_32x32x16_f16") == 240 and body.count("v_mfma_scale_f32_32x32x64_f8f6f4") == 120
    assert body.count("s_barrier") == 2 * 10 + 1          # two per K-tile body (8 drain + plain + last) + the prologue's
