#!/usr/bin/env python3
"""Scenes whose spheres overlap heavily -- the case a hierarchy cannot cull: concentric shells, a dense ball of spheres, a cube packed ten deep --
at sizes where the library walks the hierarchy without asking (1500 tree spheres and more): the pick against the plain sweep forced, 1080p, 1 pass."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402
from tools.always_list_probe import run  # noqa: E402


def base(n):
    sph = np.zeros(n + 2, api.SPHERE_DT)
    sph["rad"][0], sph["p"][0], sph["c"][0] = 1000.0, (0, -1000, 0), (.75, .75, .75)
    sph["rad"][1], sph["p"][1], sph["e"][1] = 9.0, (0, 70, 0), (14, 14, 14)
    sph["c"][2:] = 0.7
    return sph


def shells(n, rng):
    sph = base(n)
    sph["rad"][2:] = np.linspace(2.0, 20.0, n).astype(np.float32)
    sph["p"][2:] = (0, 22, 0)
    sph["refl"][2:] = api.REFR
    return sph


def ball(n, rng):
    sph = base(n)
    v = rng.normal(0, 1, (n, 3))
    v = v / np.linalg.norm(v, axis=1)[:, None] * (rng.random(n) ** (1 / 3))[:, None] * 6.0
    sph["p"][2:] = (v + np.float32([0, 12, 0])).astype(np.float32)
    sph["rad"][2:] = rng.uniform(1.0, 2.0, n).astype(np.float32)
    sph["refl"][2:] = rng.choice([api.DIFF, api.SPEC, api.REFR], n)
    return sph


def cube(n, rng):
    sph = base(n)
    sph["p"][2:] = np.stack([rng.uniform(-10, 10, n), rng.uniform(1, 21, n), rng.uniform(-10, 10, n)], 1).astype(np.float32)
    sph["rad"][2:] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    sph["refl"][2:] = rng.choice([api.DIFF, api.DIFF, api.REFR], n)
    return sph


def main():
    w, h, spp = 1920, 1080, 1
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    rng = np.random.default_rng(4)
    for name, maker in (("concentric shells", shells), ("dense ball", ball), ("packed cube", cube)):
        for n in (1600, 4000, 9000):
            sph = maker(n, rng)
            a_ms, a_k, a_px = run(sph, cam, w, h, spp, None)
            b_ms, b_k, b_px = run(sph, cam, w, h, spp, "rt_trace_parity_g", bvh_off=True)
            print(json.dumps({"scene": name, "records": int(len(sph)), "picked": a_k, "picked_ms": round(a_ms, 3), "plain_sweep": b_k, "plain_sweep_ms": round(b_ms, 3),
                              "sweep_over_picked": round(b_ms / a_ms, 3), "frames_equal": bool(np.array_equal(a_px, b_px))}), flush=True)


if __name__ == "__main__":
    main()
