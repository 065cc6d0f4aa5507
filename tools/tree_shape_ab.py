#!/usr/bin/env python3
"""Frame time and steps per ray by who shaped the hierarchy:   python tools/tree_shape_ab.py [c3,c256,c64]

shape 0 = halved on the device, 1 = by surface area on the host (any cut, three axes, partial leaves), 2 = by surface area on the
device (cuts between whole leaves, one axis per node: what updates and uploads from 1500 tree spheres get).  Same frame, interleaved
rounds in one process; pixels compared with shape 0's."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host  # noqa: E402
from ab_bench import CONFIGS  # noqa: E402
import bvh_check  # noqa: E402

lib = api.load_library(diag=True)
for name in (sys.argv[1] if len(sys.argv) > 1 else "c3").split(","):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    ctxs, times, pix, info = {}, {}, {}, {}
    for shape in (0, 1, 2):
        ctx = api.RtContext(w, h, diag=True)
        ctx._check(lib.rt_debug_set_tree_shape(ctx._h, shape))
        ctx._check(lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        b = bvh_check.read_bvh(ctx)
        info[shape] = {"leaves": b["n_leaves"], "stack_depth": b["stack_depth"], "sum_of_box_areas": round(bvh_check.sum_of_box_areas(b), 1)}
        ctxs[shape], times[shape] = ctx, []
    for r in range(6):
        for shape, ctx in ctxs.items():
            ctx.reset()
            px = ctx.render_pass(spp)
            if r == 0:
                pix[shape] = px
            elif r >= 2:                    # (the first frames price the tiles)
                times[shape].append(ctx.stats()["last_kernel_ms"])
    for shape, ctx in ctxs.items():
        ctx.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
        ctx.reset()
        ctx.render_pass(spp, copy=False)
        c = bvh_check.counters_raw(ctx)[20:30]
        st = ctx.stats()
        rays = st["closest_rays"] + st["shadow_rays"]
        rec = {"config": name, "shape": shape, "ms_median": round(statistics.median(times[shape]), 4), "ms_min": round(min(times[shape]), 4),
               "pair_steps_per_ray": round(c[1] / rays, 3), "leaf_steps_per_ray": round(c[3] / rays, 3), "same_pixels": bool(np.array_equal(pix[shape], pix[0]))}
        rec.update(info[shape])
        print(json.dumps(rec), flush=True)
        ctx.close() if hasattr(ctx, "close") else ctx.__exit__(None, None, None)
