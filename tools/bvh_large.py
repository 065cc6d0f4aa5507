#!/usr/bin/env python3
"""Scenes beyond LDS: time per frame of the walk over tables in HBM / L2, and of set_scene (host-side build beyond 16 384
spheres in the tree).  python tools/bvh_large.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracing_simple_amd import api, host
from test_gpu_bvh import _many_spheres

W, H, SPP = 1920, 1080, 4
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 20000, 65536, 262144]:
    sph, orig, target = _many_spheres(n)
    cam = host.compute_camera(orig, target, W, H)
    with api.RtContext(W, H, diag=True) as ctx:
        t0 = time.perf_counter()
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        t_set = time.perf_counter() - t0
        best = None
        for _ in range(3):
            ctx.reset()
            ctx.render_pass(SPP, copy=False)
            ms = ctx.stats()["last_kernel_ms"]
            best = ms if best is None else min(best, ms)
        st = ctx.stats()
        line = {"spheres": n, "set_scene_ms": round(t_set * 1e3, 2), "kernel": ctx.last_kernel, "ms_per_frame": round(best, 2),
                "Mray_s": round((st["samples"] + st["shadow_rays"]) / best / 1e3, 1), "pick": ctx._lib.rt_debug_bvh_pick(ctx._h)}
        if n <= 9000:
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
            ctx.reset(); ctx.render_pass(SPP, copy=False)
            line["plain_ms"] = round(ctx.stats()["last_kernel_ms"], 2)
            line["plain_kernel"] = ctx.last_kernel
    print(line, flush=True)
