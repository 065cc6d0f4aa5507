#!/usr/bin/env python3
"""Every BASELINE.json configuration at FULL size: HIP path (parity mode, through the C ABI) against
the CPU oracle on the same inputs -- pixels, colour plane, final seeds and the work counters, bit
for bit.  (The pytest suite does this at sizes the oracle finishes in seconds and checks the full
sizes through properties; this tool is the slow, exhaustive companion.)

    python tools/full_size_parity.py [c1,c2,c3,c4,c5,c16] > profiles/..._full_size_parity.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
import bench  # noqa: E402
from raytracing_simple_amd import api, host  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402

CONFIGS = dict(CONFIGS)
CONFIGS["c1"] = (lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 256, 256, 1)

for name in (sys.argv[1] if len(sys.argv) > 1 else "c1,c2,c16,c3,c5,c4").split(","):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        # three frames of the same scene and camera: the first in image order and 8x8 squares, the second with the pixels
        # priced, the third with heavy tiles first -- the frame must be the same bits each time
        same_every_frame = True
        first_px = None
        for k in range(3):
            ctx.reset()
            t0 = time.time()
            px = ctx.render_pass(spp)
            t_gpu = time.time() - t0
            if first_px is None:
                first_px = px.copy()
            same_every_frame = same_every_frame and bool(np.array_equal(px, first_px))
        got = {"pixels": px, "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats(), "kernel": ctx.last_kernel}
    cores = bench.host_cores()
    t0 = time.time()
    want = O.render(sph, cam, w, h, spp, threads=cores)
    t_cpu = time.time() - t0
    g, o = got["stats"], want["stats"]
    line = {"config": name, "spheres": int(len(sph)), "w": w, "h": h, "spp": spp, "kernel": None, "three_frames_equal": same_every_frame,
            "pixels_equal": bool(np.array_equal(got["pixels"], want["pixels"])),
            "colours_equal_bitwise": bool(np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))),
            "seeds_equal": bool(np.array_equal(got["seeds"], want["seeds"])),
            "counters_equal": (g["samples"], g["closest_rays"], g["shadow_rays"], g["sphere_tests"], g["rng_draws"]) ==
                              (o["samples"], o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"]),
            "sphere_tests": int(g["sphere_tests"]), "kernel_ms": round(g["last_kernel_ms"], 3),
            "gpu_wall_s": round(t_gpu, 3), "oracle_wall_s": round(t_cpu, 2), "oracle_threads": cores}
    line["kernel"] = got["kernel"]
    print(json.dumps(line), flush=True)
