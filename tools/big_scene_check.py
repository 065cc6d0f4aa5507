#!/usr/bin/env python3
"""The largest scenes (100 000 and RT_MAX_SPHERES = 262 144 spheres, host-built trees, tables in HBM / L2) against the oracle:
frame, colour plane, seeds and counters.  python tools/big_scene_check.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from raytracing_simple_amd import api, host
from test_gpu_bvh import _many_spheres, _same
for n in (100000, 262144):
    sph, orig, target = _many_spheres(n, seed=n)
    w, h, spp = 48, 32, 2
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp, threads=16)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        _same(got, want)
        print(n, "spheres:", ctx.last_kernel, "frame, seeds and counters equal the oracle's", flush=True)
