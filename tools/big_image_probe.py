#!/usr/bin/env python3
"""A 16384 x 8192 image (134 M pixels, indices past 2^27, 3.2 GB of device state) against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O
from raytracing_simple_amd import api, host
w, h, spp = 16384, 8192, 1
sph = host.demo_scene()
cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
t0 = time.time()
with api.RtContext(w, h) as ctx:
    ctx.set_scene(sph); ctx.set_camera(cam)
    px = ctx.render_pass(spp); st = ctx.stats(); seeds = ctx.read_seeds()
print("gpu done", round(time.time() - t0, 1), "s, kernel", round(st["last_kernel_ms"], 2), "ms", flush=True)
want = O.render(sph, cam, w, h, spp, threads=16)
print("pixels", "same" if np.array_equal(px, want["pixels"]) else "DIFF", "seeds", "same" if np.array_equal(seeds, want["seeds"]) else "DIFF",
      "counters", (st["closest_rays"], st["shadow_rays"], st["sphere_tests"]) == (want["stats"]["closest_calls"], want["stats"]["shadow_calls"], want["stats"]["sphere_tests"]))
