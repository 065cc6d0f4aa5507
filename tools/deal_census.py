#!/usr/bin/env python3
"""Census of the hierarchy walk (rt_trace_parity_pairs_census) with and without the deal of pixels by cost: python tools/deal_census.py [c3]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
import bvh_check
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
maker, w, h, spp = CONFIGS[name]
sph, orig, target = maker()
cam = host.compute_camera(orig, target, w, h)
lib = api.load_library(diag=True)
for deal in (0, 1):
    with api.RtContext(w, h, diag=True) as ctx:
        lib.rt_debug_set_pixel_deal(ctx._h, deal)
        ctx._check(lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph); ctx.set_camera(cam)
        for _ in range(3):
            ctx.reset(); ctx.render_pass(spp, copy=False)
        ms = ctx.stats()["last_kernel_ms"]
        ctx.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
        ctx.reset(); ctx.render_pass(spp, copy=False)
        c = bvh_check.counters_raw(ctx)[20:30]
        st = ctx.stats()
        print(json.dumps({"config": name, "deal": deal, "product_ms": round(ms, 3), "census_ms": round(st["last_kernel_ms"], 3),
                          "lanes_per_pair_step": round(c[1] / max(c[0], 1), 1), "lanes_per_leaf_step": round(c[3] / max(c[2], 1), 1),
                          "lanes_per_shade": round(c[5] / max(c[4], 1), 1), "wave_pair_steps": c[0], "wave_leaf_steps": c[2], "shade_phases": c[4],
                          "trips": c[8], "pair_steps_per_trip": round(c[0] / max(c[8], 1), 1)}), flush=True)
