#!/usr/bin/env python3
"""Census of the hierarchy walk: what the wavefronts of one frame execute, step by step.

    python tools/walk_census.py [c3,c256,...]

Per configuration one JSON line: wave-level pair / leaf steps and shade phases, lanes taking part in each, loop trips,
the clock shares of walk and shading, and how many lanes the leaf and pair steps run with."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host  # noqa: E402
from ab_bench import CONFIGS  # noqa: E402
import bvh_check  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "c3").split(",")
inst = sys.argv[2] if len(sys.argv) > 2 else "rt_trace_parity_pairs_census"
product = inst.replace("_census", "")
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # rt_debug_set_walk: lanes that may be left walking when a trip's walk phase ends
lib = api.load_library(diag=True)
for name in names:
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        if tail:
            ctx._check(lib.rt_debug_set_walk(ctx._h, tail, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.set_mode(api.instance_mode(product))
        for _ in range(3):
            ctx.reset()
            ctx.render_pass(spp, copy=False)
        ms = ctx.stats()["last_kernel_ms"]
        ctx.set_mode(api.instance_mode(inst))
        ctx.reset()
        ctx.render_pass(spp, copy=False)
        raw = bvh_check.counters_raw(ctx)
        st = ctx.stats()
    c, hs = raw[20:30], raw[8:18]
    rec = {"config": name, "instance": inst, "tail_lanes": tail, "product_ms": round(ms, 3), "census_ms": round(st["last_kernel_ms"], 3),
           "rays": st["closest_rays"] + st["shadow_rays"],
           "wave_pair_steps": c[0], "lanes_per_pair_step": round(c[1] / max(c[0], 1), 1),
           "wave_leaf_steps": c[2], "lanes_per_leaf_step": round(c[3] / max(c[2], 1), 1),
           "shade_phases": c[4], "lanes_per_shade": round(c[5] / max(c[4], 1), 1), "trips": c[8],
           "pair_steps_per_trip": round(c[0] / max(c[8], 1), 2), "leaf_steps_per_trip": round(c[2] / max(c[8], 1), 2),
           "pair_steps_per_ray": round(c[1] / max(st["closest_rays"] + st["shadow_rays"], 1), 2),
           "leaf_steps_per_ray": round(c[3] / max(st["closest_rays"] + st["shadow_rays"], 1), 2),
           "clock_share_walk": round(c[6] / max(c[6] + c[7], 1), 3), "always_tests": c[9]}
    rec.update({"leaf_steps_by_lanes_1_8_16_32_64": hs[0:4], "pair_steps_by_lanes_1_8_16_32_64": hs[4:8]})
    # inside a leaf step: spheres (of 8) whose discriminant is non-negative per lane, the largest such count in the wavefront, root halves the
    # wavefront executes (some lane's sphere k has one), hits, hits nearer than the best so far
    rec.update({"nonneg_per_lane_leaf": round(raw[16] / max(c[3], 1), 2), "largest_nonneg_per_leaf_step": round(raw[17] / max(c[2], 1), 2),
                "root_halves_per_leaf_step": round(raw[18] / max(c[2], 1), 2), "hits_per_lane_leaf": round(raw[19] / max(c[3], 1), 2),
                "nearer_per_lane_leaf": round(raw[31] / max(c[3], 1), 2)})
    # more rays than lanes, emulated on the walk phases this frame executed (rt_walk.inc.h, census instance): lane utilisation of the walk
    # phase as executed, and if two (four) lanes' rays of every trip were walked by ONE lane one after the other, switching for free
    m1, tot, m2, m4 = raw[0:4]
    rec["rays_per_lane_emulation"] = {"lane_utilisation_as_executed": round(tot / max(64 * m1, 1), 3),
                                      "two_rays_per_lane": round(tot / max(32 * m2, 1), 3), "four_rays_per_lane": round(tot / max(16 * m4, 1), 3),
                                      "walk_phase_length_vs_executed": {"two": round(2 * m1 / max(m2, 1), 3), "four": round(4 * m1 / max(m4, 1), 3)},
                                      "note": "steps in pair-step units (a leaf step = 4); an upper bound on what a kernel with that many rays per lane "
                                              "could reach: the switch between a lane's rays costs nothing here and occupancy is not charged"}
    print(json.dumps(rec), flush=True)
