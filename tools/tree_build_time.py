#!/usr/bin/env python3
"""Wall time of rt_set_scene (upload + tables + hierarchy) by scene size, device build (fixed shape) against the host build by
surface area.  python tools/tree_build_time.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes
lib = api.load_library(diag=True)
for n in (256, 1024, 4096, 8192, 16384):
    row = {"spheres": n}
    for by_area in (0, 1):
        with api.RtContext(64, 64, diag=True) as ctx:
            ctx._check(lib.rt_debug_set_tree_shape(ctx._h, by_area))
            a, _, _ = scenes.random_spheres(n)
            b = a.copy(); b["c"][0] *= 0.5
            ctx.set_scene(a); ctx.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 64, 64))
            ts = []
            for k in range(4):
                t0 = time.perf_counter(); ctx.set_scene(b if k % 2 == 0 else a); ctx.sync() if hasattr(ctx, 'sync') else ctx.render_pass(1, copy=False); ts.append(time.perf_counter() - t0)
            row["by_area_ms" if by_area else "device_ms"] = round(min(ts) * 1e3, 3)
    print(json.dumps(row), flush=True)
