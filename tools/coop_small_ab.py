#!/usr/bin/env python3
"""Cooperative any-hit below its 12-sphere threshold, gated on how many shadow rays are pending in the wavefront
(rt_debug_set_coop_min(min | kmax << 24)): python tools/coop_small_ab.py [c2,c9,c16]   kernel ms, same frame again / unseen passes"""
import json, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host
from ab_bench import CONFIGS
lib = api.load_library(diag=True)
for name in (sys.argv[1] if len(sys.argv) > 1 else "c2,c9,c16").split(","):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    base = None
    for label, knob in (("library", None), ("plain", 1 << 20), ("coop", 1), ("coop k<=8", 1 | 8 << 24), ("coop k<=12", 1 | 12 << 24), ("coop k<=16", 1 | 16 << 24), ("coop k<=24", 1 | 24 << 24)):
        with api.RtContext(w, h, diag=True) as ctx:
            if knob is not None:
                ctx._check(lib.rt_debug_set_coop_min(ctx._h, knob))
            ctx.set_scene(sph); ctx.set_camera(cam)
            best, unseen = None, []
            for k in range(8):
                ctx.reset(); px = ctx.render_pass(spp)
                if k >= 3:
                    ms = ctx.stats()["last_kernel_ms"]; best = ms if best is None else min(best, ms)
            for _ in range(6):
                ctx.render_pass(spp, copy=False); unseen.append(ctx.stats()["last_kernel_ms"])
            if base is None: base = px
            print(json.dumps({"config": name, "any_hit": label, "kernel": ctx.last_kernel, "ms_same_frame": round(best, 3), "ms_unseen_passes": round(statistics.median(unseen), 3),
                              "same_frame": bool(np.array_equal(px, base))}), flush=True)
