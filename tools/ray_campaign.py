#!/usr/bin/env python3
"""Many adversarial rays (tests/test_gpu_bvh.py _adversarial_rays) through the hierarchy walk and the plain sweep on a
set of scenes: python tools/ray_campaign.py [RAYS_PER_SCENE] -- prints the number of rays on which the two differ."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracing_simple_amd import api, scenes  # noqa: E402
from test_gpu_bvh import _adversarial, _adversarial_rays, _many_spheres  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
makers = {"random_1024": lambda: scenes.random_spheres(1024)[0], "random_300": lambda: scenes.random_spheres(300)[0],
          "mirror_box_120": lambda: scenes.mirror_box(120)[0], "mirror_box_700": lambda: scenes.mirror_box(700)[0],
          "slab_2500": lambda: _many_spheres(2500)[0], "demo_plus_40": lambda: scenes.demo_plus(40)[0]}
for k in range(6):
    makers[f"adversarial_{k}"] = (lambda k=k: _adversarial(k)[0])
total, bad = 0, 0
for name, mk in makers.items():
    sph = api.as_spheres(mk())
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 32))
    rays = _adversarial_rays(sph, rng, n_rays)
    with api.RtContext(32, 32, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 152 * 1024))
        ctx.set_scene(sph)
        out = np.zeros((len(rays), 4), np.uint32)
        ctx._check(ctx._lib.rt_debug_walk_rays(ctx._h, rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p)))
    differ = int(((out[:, 0] != out[:, 2]) | (out[:, 1] != out[:, 3])).sum())
    hits = int((out[0::2, 0] != 0xffffffff).sum())
    blocked = int((out[1::2, 0] < len(sph)).sum())
    print({"scene": name, "spheres": len(sph), "rays": len(rays), "closest_hits": hits, "shadow_blocked": blocked, "differ": differ}, flush=True)
    total += len(rays)
    bad += differ
print({"rays": total, "differ": bad})
