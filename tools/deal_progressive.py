#!/usr/bin/env python3
"""Does the deal of pixels by cost (and the heavy-first order) owe its gain to bench.py rendering the SAME frame again?  Times
launches of 64 passes (a) after a reset, i.e. the very frame the costs were measured on, and (b) progressively, i.e. passes the
costs have never seen (different random numbers), with the deal on and off.  python tools/deal_progressive.py [c2|c16|c3]"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
lib = api.load_library(diag=True)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c16", "c3"]):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    for deal in [int(v, 0) for v in os.environ.get('RT_DEALS', '0,32').split(',')]:
        with api.RtContext(w, h, diag=True) as ctx:
            ctx._check(lib.rt_debug_set_pixel_deal(ctx._h, deal))
            ctx.set_scene(sph); ctx.set_camera(cam)
            for _ in range(4):                          # costs, deal, order
                ctx.reset(); ctx.render_pass(spp, copy=False)
            same, fresh = [], []
            for _ in range(6):
                ctx.reset(); ctx.render_pass(spp, copy=False)
                same.append(ctx.stats()["last_kernel_ms"])
                for _ in range(3):                      # passes spp .. 4 spp - 1 of the same image: never seen before
                    ctx.render_pass(spp, copy=False)
                    fresh.append(ctx.stats()["last_kernel_ms"])
            print(json.dumps({"config": name, "deal_rows": deal, "ms_same_frame_again": round(statistics.median(same), 4),
                              "ms_passes_not_seen_before": round(statistics.median(fresh), 4)}), flush=True)
