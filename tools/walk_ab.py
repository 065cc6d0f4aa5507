#!/usr/bin/env python3
"""In-process A/B of forms of the hierarchy walk on one configuration, frames compared bit for bit with the first arm's.

    python tools/walk_ab.py c3 base tail=4 tail=8 tail=8,gate=24 l2 inst=rt_trace_parity_pairs_planes [--unseen]

An arm is a comma list of settings: tail=N (walk phase of a trip ends once <= N lanes still walk), gate=N (ready lanes that make
a wavefront shade), round=N (pair steps in a row), l2 (tables read from HBM / L2: the ..._pairs_g instance), inst=<kernel symbol>
(an instance of the diagnostics library), base (the library's defaults, hierarchy forced).  Rounds are interleaved; the figure
is the median kernel time of the same frame rendered again, or with --unseen of launches continuing the running image."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402


def parse(arm):
    out = {"tail": 0, "gate": 16, "round": 4, "l2": False, "inst": None}
    for tok in arm.split(","):
        if tok in ("base", ""):
            continue
        if tok == "l2":
            out["l2"] = True
            continue
        k, v = tok.split("=")
        out[k] = v if k == "inst" else int(v)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    unseen = "--unseen" in sys.argv
    rounds = 5
    cname, arms = args[0], args[1:] or ["base"]
    maker, w, h, spp = CONFIGS[cname]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    lib = api.load_library(diag=True)
    ctxs = {}
    for arm in arms:                    # one context per arm: each keeps its own tile order warm
        a = parse(arm)
        c = api.RtContext(w, h, diag=True)
        c._check(lib.rt_debug_set_walk(c._h, a["tail"], a["gate"], 1))
        c._check(lib.rt_debug_set_walk_round(c._h, a["round"]))
        if a["l2"]:
            c._check(lib.rt_debug_set_bvh(c._h, 56, 1024))
        c.set_scene(sph)
        c.set_camera(cam)
        if a["inst"]:
            c.set_mode(api.instance_mode(a["inst"]))
        ctxs[arm] = c
    times = {arm: [] for arm in arms}
    pix, kern = {}, {}
    for r in range(rounds + 2):
        for arm in arms:
            c = ctxs[arm]
            c.reset()
            px = c.render_pass(spp)
            kern[arm] = c.last_kernel
            if r < 2:
                pix[arm] = px
            elif unseen:
                for _ in range(3):
                    c.render_pass(spp, copy=False)
                    times[arm].append(c.stats()["last_kernel_ms"])
            else:
                times[arm].append(c.stats()["last_kernel_ms"])
    base = statistics.median(times[arms[0]])
    for arm in arms:
        t = statistics.median(times[arm])
        print(json.dumps({"config": cname, "arm": arm, "kernel": kern[arm], "ms_median": round(t, 4), "ms_min": round(min(times[arm]), 4),
                          "vs_first": round(t / base, 4), "unseen_passes": unseen, "same_frame": bool(np.array_equal(pix[arm], pix[arms[0]])),
                          "build_id": api.build_id(diag=True)}), flush=True)
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
