#!/usr/bin/env python3
"""The hierarchy walk's shading gate and pair steps in a row (rt_debug_set_walk, rt_debug_set_walk_round)
with the tile order warm:  python tools/walk_sweep.py [c3|c256|...] -- frame time, min of 4, frames compared with the first setting's."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
maker, w, h, spp = CONFIGS[name]
sph, orig, target = maker()
cam = host.compute_camera(orig, target, w, h)
lib = api.load_library(diag=True)
base = None
grid = [(16, 4), (16, 3), (16, 5), (24, 4), (32, 4), (48, 4), (32, 5), (8, 4)]
for gate, rnd in grid:
    steps = 0
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(lib.rt_debug_set_walk(ctx._h, steps, gate, 1))
        ctx._check(lib.rt_debug_set_walk_round(ctx._h, rnd))
        ctx.set_scene(sph); ctx.set_camera(cam)
        best = None
        for k in range(7):
            ctx.reset()
            px = ctx.render_pass(spp)
            if k >= 3:
                ms = ctx.stats()["last_kernel_ms"]
                best = ms if best is None else min(best, ms)
        if base is None:
            base = px
        print(json.dumps({"config": name, "gate": gate, "round": rnd, "ms": round(best, 3), "same_frame": bool(np.array_equal(px, base))}), flush=True)
