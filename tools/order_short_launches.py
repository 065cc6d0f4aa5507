#!/usr/bin/env python3
"""What the heavy-first tile order is worth on SHORT launches (1 / 2 / 4 passes per launch: the adapter's display regime), once a long
launch has built it: per scene, kernel ms of short launches with the order against natural tile order.  profiles/r06_order_short_launches.jsonl"""
import sys, os, json, statistics
sys.path.insert(0, '/root/repo')
from raytracing_simple_amd import api, host, scenes
from tools import reference_scenes
lib = api.load_library(diag=True)
LONG_FIRST = "--long-first" in sys.argv      # three 16-pass frames first (a long launch builds the order); default: short launches only -- the order comes from their own window
work = [("complex 800x600", reference_scenes.load_scene("complex"), 800, 600), ("c3 1080p", scenes.random_spheres(1024), 1920, 1080),
        ("r8192 1080p", scenes.random_spheres(8192), 1920, 1080), ("cornell 800x600", reference_scenes.load_scene("cornell"), 800, 600),
        ("c5 1080p", scenes.mirror_box(64), 1920, 1080), ("demo 1080p", (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 1920, 1080)]
for name, (sph, o, t), w, h in work:
    cam = host.compute_camera(o, t, w, h)
    for passes in (1, 2, 4):
        res = {}
        for order in (1, 0):
            with api.RtContext(w, h, diag=True) as c:
                lib.rt_debug_set_tile_order(c._h, order)        # (before anything is rendered: moving the knob later drops the order)
                c.set_scene(sph); c.set_camera(cam)
                if LONG_FIRST:
                    for _ in range(3):
                        c.reset(); c.render_pass(16, copy=False)   # costs, then the order (a launch of 8 passes or more sorts it)
                    c.reset()
                ms = []
                for k in range(72):                                # (short launches only: the window of 16 passes' worth of costs, the sort, then the order)
                    c.render_pass(passes, copy=False)
                    if k >= 40: ms.append(c.stats()["last_kernel_ms"])
                res[order] = statistics.median(ms)
        print(json.dumps({"scene": name, "order_from": "a 16-pass launch first" if LONG_FIRST else "the short launches' own window of costs", "passes_per_launch": passes, "ms_heavy_first": round(res[1], 4), "ms_natural": round(res[0], 4), "natural_over_heavy_first": round(res[0] / res[1], 3)}), flush=True)
