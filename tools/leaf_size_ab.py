#!/usr/bin/env python3
"""Spheres per leaf of the hierarchy, 8 (shipped) against 4: a second diagnostics library built with -DRT_BVH_LEAF=4 into
raytracing_simple_amd/csrc/_obj/leaf4/; tools/leaf_size_ab.sh times both on the GPU box.   python tools/leaf_size_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import _build as B
B.build()
cc = B.hipcc()
d = os.path.join(B.OBJ, "leaf4")
os.makedirs(d, exist_ok=True)
objs = []
for src, extra in B.UNITS:
    op = os.path.join(d, src + ".o")
    inc = ["-I" + B.OBJ] if src == "rt_build_id.cpp" else []
    subprocess.run([cc] + B.COMMON + ["-DRT_DIAGNOSTICS=1", "-DRT_BVH_LEAF=4"] + extra + inc + ["-c", os.path.join(B.CSRC, src), "-o", op], check=True)
    objs.append(op)
out = os.path.join(d, "librt_hip_diag.so")
subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-Wl,--version-script=" + os.path.join(B.OBJ, "exports.diag.map")], check=True)
print(out)
