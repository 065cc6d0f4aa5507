#!/usr/bin/env python3
"""Quick timing of one form of the walk on the open scenes: python tools/bvh_quick.py [FORM] [STEPS] [GATE]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import host, scenes
import bvh_check
form = int(sys.argv[1]) if len(sys.argv) > 1 else 1       # 1 = the hierarchy forced, 0 = the measured choice
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gate = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rounds = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
lds_limit = int(os.environ.get("RT_BVH_LDS_LIMIT", "0"))       # A/B: tables staged in LDS only below this many bytes
out = {}
for name, n, spp in (("1024", 1024, 16), ("512", 512, 16), ("256", 256, 16), ("2048", 2048, 8), ("box256", -256, 16)):
    sph, orig, target = scenes.random_spheres(n) if n > 0 else scenes.mirror_box(-n)
    cam = host.compute_camera(orig, target, 1920, 1080)
    t0, px0, _ = bvh_check.timed(sph, cam, 1920, 1080, spp, 0, reps=2)
    res = {}
    for r in rounds:
        t, px, st = bvh_check.timed(sph, cam, 1920, 1080, spp, 1, reps=3, walk=(steps, gate, form), ratio=r, lds_limit=lds_limit)
        res[r] = round(t, 2) if np.array_equal(px, px0) else "DIFFERENT"
    out[name] = (res, "plain", round(t0, 2))
print("quick form", form, steps, gate, out)
