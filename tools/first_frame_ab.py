#!/usr/bin/env python3
"""A new scene's FIRST frame -- what rt_render(scene, cam, out, w, h, spp) is -- under the three ways of ordering its tiles:

    python tools/first_frame_ab.py [c2,c16,c3,c5,...] > profiles/rNN_first_frame_ab.jsonl

  priced    four of the frame's passes first, in image order, to price the tiles; the rest heavy first (the library's way)
  natural   image order
(Round 5 also measured a GUESS from the scene -- the materials in front of each tile -- in the place of the pricing launch: level with it
on C2 / 16 spheres / C4 / C5, -2 % on C3, +1..4 % on 64 / 256 scattered spheres: profiles/r05_first_frame_guess_ab.jsonl, code at 2980596.)
Each figure is the device time between the events around everything the blocking frame launched, on a fresh context (GPU warm),
median of 5; `steady` is the same frame rendered again with measured costs.  Frames are compared bit for bit."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402

lib = api.load_library(diag=True)
ARMS = {"priced": 1, "natural": 0}
for cname in (sys.argv[1] if len(sys.argv) > 1 else "c2,c16,c3,c5").split(","):
    maker, w, h, spp = CONFIGS[cname]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    ms = {a: [] for a in ARMS}
    launches, pix, steady = {}, {}, []
    for r in range(6):
        for arm, knob in ARMS.items():
            with api.RtContext(w, h, diag=True) as c:
                c._check(lib.rt_debug_set_tile_order(c._h, knob))
                c.set_camera(cam)
                c.set_scene(sph)
                px = c.render_pass(spp)
                st = c.stats()
                if r > 0:
                    ms[arm].append(st["last_kernel_ms"])
                launches[arm], pix[arm] = int(st["launches"]), px
                if arm == "priced" and r > 0:
                    for _ in range(6):              # (the order settles over a few frames: fewer, and `steady` is not)
                        c.reset()
                        c.render_pass(spp, copy=False)
                    steady.append(c.stats()["last_kernel_ms"])
    # the same first frame on a context that has just rendered ANOTHER scene (buffers, streams and code warm; only the scene is new):
    # what a host that keeps its context -- or calls rt_render again at the same size -- pays for a new scene
    other = host.demo_scene() if len(sph) != 6 else CONFIGS["c16"][0]()[0]
    warm = {a: [] for a in ("priced",)}
    for arm in warm:
        with api.RtContext(w, h, diag=True) as c:
            c._check(lib.rt_debug_set_tile_order(c._h, ARMS[arm]))
            c.set_camera(cam)
            for r in range(6):
                c.set_scene(other)
                c.reset()
                c.render_pass(spp, copy=False)
                c.set_scene(sph)
                c.reset()
                px = c.render_pass(spp)
                assert np.array_equal(px, pix["natural"])
                if r > 0:
                    warm[arm].append(c.stats()["last_kernel_ms"])
    rec = {"config": cname, "spheres": int(len(sph)), "spp": spp, "steady_ms": round(statistics.median(steady), 4), "build_id": api.build_id(diag=True)}
    for arm in ARMS:
        t = statistics.median(ms[arm])
        rec[arm] = {"first_frame_ms": round(t, 4), "vs_steady": round(t / rec["steady_ms"], 3), "launches": launches[arm],
                    "same_frame": bool(np.array_equal(pix[arm], pix["natural"]))}
    for arm in warm:
        rec[arm]["first_frame_ms_on_a_warm_context"] = round(statistics.median(warm[arm]), 4)
    print(json.dumps(rec), flush=True)
