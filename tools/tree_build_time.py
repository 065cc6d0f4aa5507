#!/usr/bin/env python3
"""What a new hierarchy costs, by scene size and by who shapes it:

    python tools/tree_build_time.py

  set_scene_ms   wall time of rt_set_scene + one 64x64 pass (upload, tables, hierarchy, the pass behind them), with the tree's shape
                 0 = halved on the device, 1 = the library's choice (the host by surface area below 1500 tree spheres, the device by
                 surface area up to 4096, halved beyond), 2 = the device by surface area wherever it can
  host_ms        host time of the rt_set_scene call alone (what the caller is held for)
  update_ms      rt_update_spheres_async of the whole scene + the pass, default shape (the device by surface area up to 4096)
  update_host_ms host time of the update call alone"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402

lib = api.load_library(diag=True)
cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 64, 64)
for n in (256, 1024, 2048, 4096, 8192, 16384):
    row = {"spheres": n}
    a, _, _ = scenes.random_spheres(n)
    a = api.as_spheres(a)
    b = a.copy()
    b["c"][0] *= 0.5
    for by_area in (0, 1, 2):
        with api.RtContext(64, 64, diag=True) as ctx:
            ctx._check(lib.rt_debug_set_tree_shape(ctx._h, by_area))
            ctx.set_scene(a)
            ctx.set_camera(cam)
            ctx.render_pass(1, copy=False)
            total, held = [], []
            for k in range(5):
                t0 = time.perf_counter()
                ctx.set_scene(b if k % 2 == 0 else a)
                t1 = time.perf_counter()
                ctx.render_pass(1, copy=False)
                total.append(time.perf_counter() - t0)
                held.append(t1 - t0)
            row[f"set_scene_ms_{by_area}"] = round(min(total) * 1e3, 3)
            row[f"host_ms_{by_area}"] = round(min(held) * 1e3, 3)
            if by_area == 1:
                total, held = [], []
                for k in range(5):
                    t0 = time.perf_counter()
                    ctx.update_spheres(0, b if k % 2 == 0 else a)
                    t1 = time.perf_counter()
                    ctx.render_pass(1, copy=False)
                    total.append(time.perf_counter() - t0)
                    held.append(t1 - t0)
                row["update_ms"] = round(min(total) * 1e3, 3)
                row["update_host_ms"] = round(min(held) * 1e3, 3)
    print(json.dumps(row), flush=True)
