#!/usr/bin/env python3
"""Wall time of the one-shot entry point rt_render() (context, seeds, launch, readback, destroy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
for (w, h, spp) in [(256, 256, 1), (1920, 1080, 64), (1920, 1080, 64), (1920, 1080, 64), (800, 600, 1), (800, 600, 1)]:
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    t0 = time.perf_counter()
    px = api.render(host.demo_scene(), cam, w, h, spp)
    print(f"rt_render {w}x{h} x {spp} spp: {(time.perf_counter() - t0) * 1e3:.2f} ms wall", flush=True)
