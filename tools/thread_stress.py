#!/usr/bin/env python3
"""Contexts driven from concurrent host threads, many rounds; prints what differs (if anything)."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from raytracing_simple_amd import api, host, scenes
jobs = [(scenes.demo_plus(16), 160, 96, 6), ((host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 200, 120, 5),
        (scenes.random_spheres(96), 96, 64, 4), (scenes.mirror_box(64), 64, 64, 3)] * 2

def one(job):
    (sph, orig, target), w, h, spp = job
    cam = host.compute_camera(orig, target, w, h)
    out = []
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        for _ in range(3):
            ctx.reset(); out.append((ctx.render_pass(spp), ctx.stats()))
    return out

serial = [one(j) for j in jobs[:4]]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for r in range(rounds):
    try:
        with ThreadPoolExecutor(max_workers=int(os.environ.get("RT_STRESS_THREADS", "8"))) as pool:
            threaded = list(pool.map(one, jobs))
    except Exception:
        bad += 1
        print("round", r, "EXCEPTION"); traceback.print_exc(); continue
    for k, got in enumerate(threaded):
        for f, (frame, st) in enumerate(got):
            ref = serial[k % 4][0][0]
            if not np.array_equal(frame, ref):
                bad += 1
                d = np.flatnonzero(frame != ref)
                print("round", r, "job", k, "frame", f, "differs in", d.size, "pixels; first", d[:6], "stats", st["samples"], st["closest_rays"],
                      "ref stats", serial[k % 4][0][1]["samples"], serial[k % 4][0][1]["closest_rays"], flush=True)
print("thread stress:", rounds, "rounds,", bad, "problems")
