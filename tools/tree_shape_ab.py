#!/usr/bin/env python3
"""The hierarchy's shape: the device build's (leaf ranges halved) against the host build's of a full scene upload (cuts chosen by
surface area, rt_bvh.hip build_on_host_sah) -- kernel ms per launch on the same frame rendered again and on passes not rendered
before, leaves, stack depth, the library's choice.  python tools/tree_shape_ab.py [c3,c256,c64,c5]"""
import ctypes as C, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
import bvh_check
lib = api.load_library(diag=True)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c3", "c256", "c64"]):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    for by_area in (0, 1):
        with api.RtContext(w, h, diag=True) as ctx:
            ctx._check(lib.rt_debug_set_tree_shape(ctx._h, by_area))
            ctx.set_scene(sph); ctx.set_camera(cam)
            for _ in range(4):
                ctx.reset(); ctx.render_pass(spp, copy=False)
            same, fresh = [], []
            for _ in range(6):
                ctx.reset(); ctx.render_pass(spp, copy=False)
                same.append(ctx.stats()["last_kernel_ms"])
                for _ in range(3):
                    ctx.render_pass(spp, copy=False)
                    fresh.append(ctx.stats()["last_kernel_ms"])
            b = bvh_check.read_bvh(ctx)
            print(json.dumps({"config": name, "shape": "by surface area (host)" if by_area else "halved (device)", "kernel": ctx.last_kernel,
                              "leaves": b["n_leaves"] if b else None, "stack_depth": b["stack_depth"] if b else None,
                              "ms_same_frame_again": round(statistics.median(same), 4), "ms_passes_not_seen_before": round(statistics.median(fresh), 4),
                              "choice": ctx.scene_choice()}), flush=True)
