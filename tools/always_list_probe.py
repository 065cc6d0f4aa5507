#!/usr/bin/env python3
"""Scenes of two size classes -- thousands of small spheres ("dust") and hundreds to thousands of spheres more than 16 x their median radius ("objects"),
which the hierarchy keeps outside the tree and sweeps for every ray (the always-list): the library's pick against the plain sweep through the scalar
cache forced by name, 1080p, median kernel ms, frames compared bit for bit.  `python tools/always_list_probe.py` (diagnostics library)."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402


def two_classes(n_small, n_big, seed=3):
    rng = np.random.default_rng(seed)
    n = 2 + n_small + n_big
    sph = np.zeros(n, api.SPHERE_DT)
    sph["rad"][0], sph["p"][0], sph["c"][0] = 1000.0, (0, -1000, 0), (.75, .75, .75)
    sph["rad"][1], sph["p"][1], sph["e"][1] = 9.0, (0, 70, 0), (14, 14, 14)
    s = slice(2, 2 + n_small)
    sph["rad"][s] = rng.uniform(0.02, 0.05, n_small).astype(np.float32)
    sph["p"][s] = np.stack([rng.uniform(-60, 60, n_small), rng.uniform(0.05, 12, n_small), rng.uniform(-60, 60, n_small)], 1).astype(np.float32)
    b = slice(2 + n_small, n)
    sph["rad"][b] = rng.uniform(1.0, 2.5, n_big).astype(np.float32)
    sph["p"][b] = np.stack([rng.uniform(-70, 70, n_big), sph["rad"][b], rng.uniform(-70, 70, n_big)], 1).astype(np.float32)
    sph["c"][2:] = rng.uniform(0.2, 0.9, (n - 2, 3)).astype(np.float32)
    sph["refl"][2:] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n - 2)
    return sph, host.DEMO_ORIG, host.DEMO_TARGET


def run(sph, cam, w, h, spp, inst, bvh_off=False):
    with api.RtContext(w, h, diag=True) as ctx:
        if bvh_off:
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        if inst:
            ctx.set_mode(api.instance_mode(inst))
        ts, px = [], None
        for _ in range(4):
            ctx.reset()
            px = ctx.render_pass(spp)
            ts.append(ctx.stats()["last_kernel_ms"])
        return statistics.median(ts[1:]), ctx.last_kernel, px


def main():
    w, h, spp = 1920, 1080, 1
    for n_small, n_big in ((6000, 0), (6000, 100), (6000, 500), (6000, 2000), (6000, 4000), (2000, 2000), (20000, 4000), (1000, 600)):
        sph, orig, target = two_classes(n_small, n_big)
        cam = host.compute_camera(orig, target, w, h)
        a_ms, a_k, a_px = run(sph, cam, w, h, spp, None)
        b_ms, b_k, b_px = run(sph, cam, w, h, spp, "rt_trace_parity_g", bvh_off=True)
        print(json.dumps({"small": n_small, "large": n_big, "records": int(len(sph)), "picked": a_k, "picked_ms": round(a_ms, 3), "plain_sweep": b_k,
                          "plain_sweep_ms": round(b_ms, 3), "sweep_over_picked": round(b_ms / a_ms, 3), "frames_equal": bool(np.array_equal(a_px, b_px))}), flush=True)


if __name__ == "__main__":
    main()
