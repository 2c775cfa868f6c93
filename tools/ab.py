#!/usr/bin/env python
"""A/B timing of kernel variants.  Box-to-box noise between gpurun calls is 2-3 %, so variants are compared inside ONE
call:  build here      : python tools/ab.py build NAME [-DFLAG=1 ...]   -> det-sam2_amd/lib/ab_NAME.so
       run on the box  : python tools/ab.py run NAME1 NAME2 ... [--rounds 3] (alternates the builds, prints fps + stages)
       env variants    : python tools/ab.py env base name1:VAR=1 name2:VAR=2,OTHER=x ... [--rounds 3]  (default library, or
                         det-sam2_amd/lib/ab_<name>.so when it exists; the listed variables are set for that variant)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(name, flags):
    import __graft_entry__ as g
    from concurrent.futures import ThreadPoolExecutor
    out = os.path.join(g.PKG, "lib", f"ab_{name}.so")
    obj_dir = os.path.join(g.PKG, "lib", f"obj_ab_{name}")
    with ThreadPoolExecutor(max_workers=6) as ex:      # per source, with the per-source flags of the default build
        objs = list(ex.map(lambda s: g._compile_one(os.path.join(g.PKG, "csrc", s), True, obj_dir, flags), g.SOURCES))
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    g._build_torch_ops(True, out, out[:-3] + "_torch.so")     # the op library of this variant, linked against it
    print(out)


def run(names, rounds, env_mode=False):
    import __graft_entry__ as g
    for r in range(rounds):
        for spec in names:
            n, _, kv = spec.partition(":")
            env = dict(os.environ)
            lib = os.path.join(g.PKG, "lib", f"ab_{n}.so")
            if not env_mode or os.path.exists(lib):
                env["DS2_LIB"] = lib
            for item in filter(None, kv.split(",")):
                k, _, v = item.partition("=")
                env[k] = v
            o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "1", "--no-cpu-baseline", "--no-stream"],
                               env=env, capture_output=True, text=True)
            try:
                d = json.loads(o.stdout.strip().splitlines()[-1])
                gm = d.get("roofline_gemm") or {}
                print(f"{n:12s} fps {d['value']:.2f} cross {d['roofline']['avg_launch_ms']:.3f} ms  gemm {gm.get('family_ms_per_frame', 0):.3f} ms/frame "
                      f"(top {gm.get('shape')} {gm.get('achieved', 0):.0f} TF)  k_mlp256 {((d.get('by_kernel') or {}).get('k_mlp256') or {}).get('avg_launch_ms', 0):.4f} ms  {d['ms_per_step_by_stage']}", flush=True)
            except Exception:
                print(n, "FAILED", o.stderr[-400:])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        a = sys.argv[2:]
        rounds = 3
        if "--rounds" in a:
            i = a.index("--rounds")
            rounds = int(a[i + 1])
            a = a[:i] + a[i + 2:]
        run(a, rounds, env_mode=(sys.argv[1] == "env"))
