#!/usr/bin/env python3
"""Calibration record of the estimate that settles hierarchy against sweep without a launch (csrc/rt_api.hip estimate_ratio).

For each scene of three families (spheres scattered on a plane, a closed box packed with mirror / glass spheres, a cloud in
the air over a ground sphere) the library's own probe times both forms (the estimate switched off), and the tool prints that
next to the tree's surface-area sums and the ratio the estimate predicts:

    python tools/choice_calibration.py [--fit] > profiles/r04a_choice_calibration.jsonl

--fit: least squares for the four weights on the records (log of the measured ratio), printed to stderr."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402


def cloud(count, extent=60.0, height=40.0, rad=(1.0, 3.0)):
    """ground + light + (count - 2) spheres in a slab of air: rays cross many boxes"""
    rng = scenes._Mwc(0x1234567, 0x7654321)
    parts = [scenes._sphere(1000, (0, -1000, 0), (0, 0, 0), (.75, .75, .75), api.DIFF),
             scenes._sphere(7, (0, 90, 0), (12, 12, 12), (0, 0, 0), api.DIFF)]
    for i in range(count - 2):
        r = rng.uniform(*rad)
        p = (rng.uniform(-extent, extent), rng.uniform(r, height), rng.uniform(-extent, extent))
        col = (rng.uniform(.1, .9), rng.uniform(.1, .9), rng.uniform(.1, .9))
        parts.append(scenes._sphere(r, p, (0, 0, 0), col, (api.DIFF, api.SPEC, api.REFR)[i % 3]))
    return np.concatenate(parts), host.DEMO_ORIG, host.DEMO_TARGET


FAMILIES = {
    "plane": (scenes.random_spheres, [64, 80, 97, 128, 160, 200, 256, 400, 512, 800, 1024, 1400]),
    "box": (scenes.mirror_box, [64, 80, 96, 128, 160, 200, 256, 400, 600]),
    "cloud": (cloud, [64, 97, 128, 200, 256, 400, 600, 1000]),
    "demo_plus": (scenes.demo_plus, [64, 100, 160]),
}


def main():
    w, h, spp = 1920, 1080, 16
    recs = []
    for fam, (maker, counts) in FAMILIES.items():
        for n in counts:
            sph, orig, target = maker(n)
            cam = host.compute_camera(orig, target, w, h)
            with api.RtContext(w, h, diag=True) as ctx:
                ctx._check(ctx._lib.rt_debug_set_choice_estimate(ctx._h, 0))
                ctx.set_scene(sph)
                ctx.set_camera(cam)
                ctx.render_pass(spp, copy=False)
                ch = ctx.scene_choice()
                est = (C.c_double * 4)()
                have = ctx._lib.rt_debug_tree_estimate(ctx._h, est)
                cnt = (C.c_uint32 * 4)()
                ctx._check(ctx._lib.rt_debug_read_bvh(ctx._h, None, 0, cnt))
                st = ctx.stats()
            rec = {"family": fam, "n": int(len(sph)), "n_always": int(cnt[0]), "n_leaves": int(cnt[1]), "have_estimate": int(have),
                   "est_pairs": round(est[0], 3), "est_leaves": round(est[1], 3), "est_ratio": round(est[2], 3),
                   "picked": ch["picked"], "hierarchy_ms_per_pass": round(ch["hierarchy_ms_per_pass"], 4),
                   "sweep_ms_per_pass": round(ch["sweep_ms_per_pass"], 4),
                   "measured_ratio": round(ch["hierarchy_ms_per_pass"] / ch["sweep_ms_per_pass"], 3) if ch["sweep_ms_per_pass"] else None,
                   "rays_per_pass": (st["closest_rays"] + st["shadow_rays"]) // spp}
            recs.append(rec)
            print(json.dumps(rec), flush=True)
    if "--fit" in sys.argv:
        fit(recs)


def fit(recs):
    """ratio = (a P + b L + n_always + c) / (n + d): Gauss-Newton on log ratio, from the weights in the library"""
    rs = [r for r in recs if r["measured_ratio"] and r["have_estimate"]]
    P = np.array([r["est_pairs"] for r in rs]); L = np.array([r["est_leaves"] for r in rs])
    A = np.array([r["n_always"] for r in rs], float); N = np.array([r["n"] for r in rs], float)
    y = np.log(np.array([r["measured_ratio"] for r in rs]))
    x = np.array([6.0, 14.0, 30.0, 24.0])

    def model(x):
        return np.log((x[0] * P + x[1] * L + A + x[2]) / (N + x[3]))
    for _ in range(200):
        f = model(x) - y
        J = np.zeros((len(rs), 4))
        for k in range(4):
            d = np.zeros(4); d[k] = 1e-4 * max(1.0, abs(x[k]))
            J[:, k] = (model(x + d) - model(x)) / d[k]
        step = np.linalg.lstsq(J, -f, rcond=None)[0]
        x = np.maximum(x + 0.5 * step, 0.0)
    res = model(x) - y
    print("fit: kEstPair %.2f kEstLeaf %.2f kEstWalkFixed %.1f kEstSweepFixed %.1f; rms log error %.3f, worst %.3f" %
          (x[0], x[1], x[2], x[3], float(np.sqrt((res ** 2).mean())), float(np.abs(res).max())), file=sys.stderr)
    for r, e in zip(rs, res):
        print("  %-9s n %5d measured %.3f predicted %.3f" % (r["family"], r["n"], r["measured_ratio"], r["measured_ratio"] * np.exp(e)), file=sys.stderr)


if __name__ == "__main__":
    main()
