#!/usr/bin/env python3
"""C3's 1024 spheres (and 8192) moved far from the origin -- centres and camera shifted by 0, 1e4, 1e5, 1e6 along x: the walk's boxes carry a pad that
grows with |origin| (64 u (|o| + far + r): the slab arithmetic's rounding), so far from the origin the hierarchy culls less.  The pick against both forms
forced, 1080p, 2 passes."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402
from tools.choice_fuzz import run  # noqa: E402


def main():
    w, h, spp = 1920, 1080, 2
    for n in (1024, 8192):
        base, orig, target = scenes.random_spheres(n)
        for shift in (0.0, 1e4, 1e5, 1e6):
            sph = base.copy()
            sph["p"][:, 0] += np.float32(shift)
            o = (orig[0] + shift, orig[1], orig[2])
            t = (target[0] + shift, target[1], target[2])
            cam = host.compute_camera(o, t, w, h)
            res = {}
            px0 = None
            for how in ("pick", "hierarchy", "sweep"):
                ms, k, px = run(sph, cam, w, h, spp, how)
                res[how] = {"ms": round(ms, 3), "kernel": k}
                if px0 is None:
                    px0 = px
                elif not np.array_equal(px0, px):
                    res[how]["FRAME_DIFFERS"] = True
            best = min(v["ms"] for v in res.values())
            print(json.dumps({"spheres": n, "shift_x": shift, "forms": res, "worse_than_best": round(res["pick"]["ms"] / best, 2)}), flush=True)


if __name__ == "__main__":
    main()
