#!/usr/bin/env python3
"""The walk whose lanes share the rays' work (rt_walk_share.inc.h) against the shipped walk, in one process:

    python tools/walk_share_ab.py [c3,c256,c64] [--takes 1,4,8,16,32,65] [--rounds 4] [--kernel rt_trace_parity_pairs_share]

Per configuration: the library's own choice first (mode 0), then the instance with every `take` (idle lanes that start a
take-over phase; 65 = never).  Frame time = median kernel time over interleaved rounds on the same frame; pixels, colour
plane, seeds and the five counters are compared with mode 0's.  With --census one more line per take: steps per wavefront
and per lane, take-over phases, pieces taken."""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="?", default="c3")
ap.add_argument("--takes", default="1,4,8,16,32,65")
ap.add_argument("--round", type=int, default=4)
ap.add_argument("--passes", type=int, default=1, help="pairings in a row per take-over phase")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--kernel", default="rt_trace_parity_pairs_share")
ap.add_argument("--census", action="store_true")
args = ap.parse_args()
takes = [int(t) for t in args.takes.split(",")]
lib = api.load_library(diag=True)
KEYS = ("samples", "closest_rays", "shadow_rays", "sphere_tests", "rng_draws")
for name in args.configs.split(","):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        variants = [(0, None)] + [(api.instance_mode(args.kernel), t) for t in takes]
        times = {v: [] for v in variants}
        out = {}
        for r in range(args.rounds + 1):
            for v in variants:
                mode, take = v
                ctx.set_mode(mode)
                ctx._check(lib.rt_debug_set_walk_round(ctx._h, args.round | (((take or 8) | (args.passes - 1) << 8) << 8)))
                ctx.reset()
                px = ctx.render_pass(spp)
                st = ctx.stats()
                if r == 0:
                    out[v] = (px, ctx.read_colors().copy(), ctx.read_seeds().copy(), tuple(st[k] for k in KEYS), ctx.last_kernel)
                else:
                    times[v].append(st["last_kernel_ms"])
        base = out[variants[0]]
        for v in variants:
            mode, take = v
            o = out[v]
            rec = {"config": name, "kernel": o[4], "take": take, "ms_median": round(statistics.median(times[v]), 4), "ms_min": round(min(times[v]), 4),
                   "pixels_equal": bool(np.array_equal(o[0], base[0])), "colours_equal": bool(np.array_equal(o[1].view(np.uint32), base[1].view(np.uint32))),
                   "seeds_equal": bool(np.array_equal(o[2], base[2])), "counters_equal": o[3] == base[3]}
            print(json.dumps(rec), flush=True)
        if args.census:
            for take in takes:
                ctx.set_mode(api.instance_mode(args.kernel + "_census"))
                ctx._check(lib.rt_debug_set_walk_round(ctx._h, args.round | ((take | (args.passes - 1) << 8) << 8)))
                ctx.reset()
                ctx.render_pass(spp, copy=False)
                c = bvh_check.counters_raw(ctx)[20:30]
                st = ctx.stats()
                rays = st["closest_rays"] + st["shadow_rays"]
                print(json.dumps({"config": name, "census": args.kernel, "take": take, "trips": c[8], "wave_pair_steps": c[0], "lanes_per_pair_step": round(c[1] / max(c[0], 1), 1),
                                  "wave_leaf_steps": c[2], "lanes_per_leaf_step": round(c[3] / max(c[2], 1), 1), "pair_steps_per_trip": round(c[0] / max(c[8], 1), 2),
                                  "leaf_steps_per_trip": round(c[2] / max(c[8], 1), 2), "pair_steps_per_ray": round(c[1] / max(rays, 1), 2),
                                  "leaf_steps_per_ray": round(c[3] / max(rays, 1), 2), "take_phases_per_trip": round(c[4] / max(c[8], 1), 2),
                                  "pieces_taken_per_ray": round(c[5] / max(rays, 1), 3)}), flush=True)
