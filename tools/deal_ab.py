#!/usr/bin/env python3
"""In-process A/B of the deal of pixels to wavefronts by cost (rt_order_pixels_kernel) on the BASELINE configurations:
the same frame with 8x8 squares (0) and with regions of 8 / 16 / 32 rows, interleaved rounds.   python tools/deal_ab.py [c2,c16,c3,c5,c4] [rounds]"""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "c2,c16,c3,c5").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = api.load_library(diag=True)
DEALS = [int(v, 0) for v in os.environ.get("RT_DEALS", "0,8,16,32").split(",")]    # rows of a deal region (0 = no deal) | pixels of a run << 8 (0x420 = runs of 4, 32 rows)
for name in names:
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    ctxs = {}
    for deal in DEALS:
        c = api.RtContext(w, h, diag=True)
        lib.rt_debug_set_pixel_deal(c._h, deal)
        c.set_scene(sph); c.set_camera(cam)
        for _ in range(4):                                   # probes of a large scene, costs, deal, order
            c.reset(); c.render_pass(spp, copy=False)
        ctxs[deal] = c
    times = {d: [] for d in DEALS}
    px = {}
    for _ in range(rounds):
        for deal in DEALS:
            c = ctxs[deal]
            c.reset()
            px[deal] = c.render_pass(spp)
            times[deal].append(c.stats()["last_kernel_ms"])
    st = ctxs[DEALS[-1]].stats()
    rays = st["samples"] + st["shadow_rays"]
    for deal in DEALS:
        med = statistics.median(times[deal])
        print(json.dumps({"config": name, "deal": deal, "kernel": ctxs[deal].last_kernel, "ms_median": round(med, 4), "ms_min": round(min(times[deal]), 4),
                          "Gray_s": round(rays / med / 1e6, 2), "same_frame": bool(np.array_equal(px[DEALS[0]], px[deal]))}), flush=True)
    for c in ctxs.values():
        c.close()
