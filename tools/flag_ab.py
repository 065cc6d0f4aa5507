#!/usr/bin/env python3
"""Compiler-flag A/B of the render kernels: builds the diagnostics library once per flag set (only the two kernel units are
recompiled) into raytracing_simple_amd/csrc/_obj/flagab/<tag>/librt_hip_diag.so; tools/flag_ab.sh then times each on the GPU box.
    python tools/flag_ab.py            # build all variants here (hipcc cross-compiles)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import _build as B
VARIANTS = {
    "base": [],
    "maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "maxclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "iterilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "iterminreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
    "itermaxocc": ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"],
    "bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "bias100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
    "trackers": ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
    "nohighrp": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
    "prealloc": ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs=1"],
    "O2": ["-O2"],
}
def main():
    B.build()
    cc = B.hipcc()
    vs = os.path.join(B.OBJ, "exports.diag.map")
    for tag, flags in VARIANTS.items():
        d = os.path.join(B.OBJ, "flagab", tag)
        os.makedirs(d, exist_ok=True)
        objs = []
        ok = True
        for src, extra in B.UNITS:
            base_obj = os.path.join(B.OBJ, src + ".diag.o")
            if src.startswith("rt_kernel_"):
                op = os.path.join(d, src + ".o")
                cmd = [cc] + B.COMMON + ["-DRT_DIAGNOSTICS=1"] + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", op]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    print(tag, "does not compile:", r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?")
                    ok = False
                    break
                objs.append(op)
            else:
                objs.append(base_obj)
        if not ok:
            continue
        out = os.path.join(d, "librt_hip_diag.so")
        subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-Wl,--version-script=" + vs], check=True)
        print(tag, "built")
if __name__ == "__main__":
    main()
