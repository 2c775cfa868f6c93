#!/usr/bin/env python
"""Build det-sam2_amd/lib/ab_<name>.so with the assembly GEMM bodies regenerated under other generator settings (only gemm_x4g.hip is
recompiled; the other objects are the default build's):

    python tools/x4g_variant.py NAME [generator flags: nodrain nodma noread nomfma nobarrier nostore nogelu] [VAR=value: X4G_GAP=32 ...]
    on the GPU box:  DS2_LIB=det-sam2_amd/lib/ab_NAME.so python tools/x4g_check.py big 5 --nocheck
Flags give WRONG results by construction (timing ablations)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name = sys.argv[1]
flags = [a for a in sys.argv[2:] if "=" not in a]
env = dict(os.environ, **dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a))
d = os.path.join(g.PKG, "lib", f"obj_ab_{name}")
os.makedirs(d, exist_ok=True)
for cfg in ("42", "23", "23m"):
    for epi in ("e1", "e2", "e3"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_gemm_x4g.py"), os.path.join(d, f"gemm_x4g_body_{cfg}_{epi}.inc"), cfg, epi] + flags,
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode:      # (a flag that configuration does not take: its committed body)
            import shutil
            shutil.copy(os.path.join(g.PKG, "csrc", f"gemm_x4g_body_{cfg}_{epi}.inc"), os.path.join(d, f"gemm_x4g_body_{cfg}_{epi}.inc"))
            print(f"(cfg {cfg} {epi}: generator rejected the flags, committed body used)")
obj = g._compile_one(os.path.join(g.PKG, "csrc", "gemm_x4g.hip"), True, d, [f"-DX4G_INC_DIR={d}", "-Wno-inline-asm"])
objs = [os.path.join(g.OBJ_DIR, s + ".o") for s in g.SOURCES if s != "gemm_x4g.hip"] + [obj]
out = os.path.join(g.PKG, "lib", f"ab_{name}.so")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
g._build_torch_ops(True, out, out[:-3] + "_torch.so")
print(out)
