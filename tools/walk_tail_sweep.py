#!/usr/bin/env python3
"""The second form of the walk kernel: up to how many lanes is a leaf step done by the wavefront (rt_debug_set_walk_tail; one
sphere test per lane, eight lanes per ray) instead of by each lane for itself?   python tools/walk_tail_sweep.py [c3,c256] [instance]
Per setting: kernel ms of the same frame rendered again (min of 4) and of passes not rendered before (median of 6 launches),
frames compared with the first setting's."""
import json, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host
from ab_bench import CONFIGS
names = (sys.argv[1] if len(sys.argv) > 1 else "c3").split(",")
inst = sys.argv[2] if len(sys.argv) > 2 else "rt_trace_parity_pairs2"
lib = api.load_library(diag=True)
grid = [0, 4, 8, 12, 16, 24, 32]
if os.environ.get("RT_TAILS"):
    grid = [int(v) for v in os.environ["RT_TAILS"].split(",")]
for name in names:
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    base = None
    for lanes in grid:
        with api.RtContext(w, h, diag=True) as ctx:
            ctx._check(lib.rt_debug_set_walk_tail(ctx._h, lanes))
            ctx.set_scene(sph); ctx.set_camera(cam)
            ctx.set_mode(api.instance_mode(inst))
            best, unseen = None, []
            for k in range(7):
                ctx.reset()
                px = ctx.render_pass(spp)
                if k >= 3:
                    ms = ctx.stats()["last_kernel_ms"]
                    best = ms if best is None else min(best, ms)
            for _ in range(6):
                ctx.render_pass(spp, copy=False)
                unseen.append(ctx.stats()["last_kernel_ms"])
            if base is None:
                base = px
            print(json.dumps({"config": name, "instance": inst, "coop_leaf_up_to_lanes": lanes, "ms_same_frame": round(best, 3),
                              "ms_unseen_passes": round(statistics.median(unseen), 3), "same_frame": bool(np.array_equal(px, base))}), flush=True)
