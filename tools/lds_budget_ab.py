#!/usr/bin/env python3
"""Where do a hierarchy's tables belong -- all of them in LDS at fewer workgroups per CU, or the pairs in LDS and the slots in HBM / L2 at five?

    python tools/lds_budget_ab.py [--sizes 1200,1600,2048,2400,3000] [--limits 31,40,52,76] [--scn complex]

rt_launch.hip stages the whole tables while they fit `bvh_lds_limit` (31 KiB: five workgroups of four wavefronts per CU) and falls back to
rt_trace_*_pairs_m (pairs staged, slots read where they lie) beyond it.  profiles/r06_reference_scenes.jsonl showed complex.scn (1566 records)
3 % FASTER with everything in LDS at three workgroups per CU than with the library's pick: this sweep measures the limit on scattered-sphere
scenes of the sizes around it (in-process, interleaved rounds, frames compared bit for bit with the first arm's)."""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1200,1600,2048,2400,3000")
    ap.add_argument("--limits", default="31,40,52,76")
    ap.add_argument("--scn", default="complex")
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    lib = api.load_library(diag=True)
    limits = [int(v) for v in args.limits.split(",")]
    work = [("r%d" % int(n), scenes.random_spheres(int(n)), 1920, 1080) for n in args.sizes.split(",") if n]
    if args.scn:
        from tools import reference_scenes
        for name in args.scn.split(","):
            work.append(("scn:" + name, reference_scenes.load_scene(name), 800, 600))
    for name, (sph, orig, target), w, h in work:
        cam = host.compute_camera(orig, target, w, h)
        ctxs = {}
        for kib in limits:
            c = api.RtContext(w, h, diag=True)
            c._check(lib.rt_debug_set_bvh(c._h, 56, kib * 1024))
            c._check(lib.rt_debug_set_walk(c._h, 0, 0, 1))
            c.set_scene(sph)
            c.set_camera(cam)
            ctxs[kib] = c
        times, pix, kern = {k: [] for k in limits}, {}, {}
        for r in range(args.rounds + 2):
            for kib in limits:
                c = ctxs[kib]
                c.reset()
                px = c.render_pass(args.spp)
                kern[kib] = c.last_kernel
                if r < 2:
                    pix[kib] = px
                else:
                    times[kib].append(c.stats()["last_kernel_ms"])
        base = statistics.median(times[limits[0]])
        for kib in limits:
            t = statistics.median(times[kib])
            print(json.dumps({"scene": name, "spheres": int(len(sph)), "w": w, "h": h, "spp": args.spp, "lds_limit_KiB": kib, "kernel": kern[kib],
                              "ms_median": round(t, 4), "vs_first": round(t / base, 4), "same_frame": bool(np.array_equal(pix[kib], pix[limits[0]])),
                              "build_id": api.build_id(diag=True)}), flush=True)
        for c in ctxs.values():
            c.close()


if __name__ == "__main__":
    main()
