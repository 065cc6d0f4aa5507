#!/usr/bin/env python3
"""Many small renders with the census instance (mode 105); run under
rocprofv3 --pmc SQ_INSTS_VALU, then tools/regress_fit.py joins PMC rows with the census lines."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_simple_amd import api, host, scenes
cases = []
for name, maker in (("demo", lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET)),
                    ("plus16", lambda: scenes.demo_plus(16)), ("box", lambda: scenes.mirror_box(24))):
    sph, orig, target = maker()
    for (w, h) in ((64, 64), (128, 32), (32, 128), (96, 96)):
        for shift in (0.0, 15.0, -25.0):
            o = (orig[0] + shift, orig[1] + 0.3 * shift, orig[2])
            cases.append((name, sph, o, target, w, h))
for name, sph, o, t, w, h in cases:
    cam = host.compute_camera(o, t, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam); ctx.set_mode(105)
        ctx.render_pass(8, copy=False)
        st = ctx.stats()
        buf = (C.c_ulonglong * 24)()
        api.load_library(diag=True).rt_debug_counters(ctx._h, buf)
        v = list(buf)
        print("CENSUS " + json.dumps({"name": name, "w": w, "h": h, "n": len(sph), "execs": [c >> 32 for c in v[:10]],
                                      "lanes": [c & 0xFFFFFFFF for c in v[:10]], "roots_c": v[10], "roots_s": v[11],
                                      "tests": st["sphere_tests"], "closest": st["closest_rays"], "shadow": st["shadow_rays"]}), flush=True)
