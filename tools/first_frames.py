#!/usr/bin/env python3
"""The first frames of a scene: frame time and launches per frame (a long launch without tile costs renders 4 passes first to
price the tiles, rt_api.hip launch_priced), against the same frame in image order with 8x8 squares (what a first frame was
before) and the steady state.  python tools/first_frames.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
lib = api.load_library(diag=True)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c16", "c3", "c5"]):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        for k in range(6):                                   # the GPU warm (clocks, code, caches) before anything is compared
            ctx.reset(); ctx.render_pass(spp, copy=False)
        other = sph.copy()
        other["c"][0] = other["c"][0] * 0.5                  # another scene (one colour differs): costs, deal and order are dropped
        ctx.set_scene(other)
        ms, launches = [], []
        for k in range(5):
            ctx.reset(); ctx.render_pass(spp, copy=False)
            st = ctx.stats(); ms.append(round(st["last_kernel_ms"], 3)); launches.append(st["launches"])
        ctx._check(lib.rt_debug_set_tile_order(ctx._h, 0))
        ctx._check(lib.rt_debug_set_pixel_deal(ctx._h, 0))
        plain = []
        for k in range(3):
            ctx.reset(); ctx.render_pass(spp, copy=False)
            plain.append(round(ctx.stats()["last_kernel_ms"], 3))
        print(json.dumps({"config": name, "frame_ms": ms, "launches_per_frame": launches, "image_order_squares_ms": plain}), flush=True)
