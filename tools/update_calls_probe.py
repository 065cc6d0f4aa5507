#!/usr/bin/env python3
"""What a frame costs when K scattered spheres move by K calls of rt_update_spheres_async against ONE call over the whole range: C3's 1024 spheres
and an 8192-sphere scene, 1080p, wall clock per frame (update calls + reset + launch + wait) over 20 frames."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402


def main():
    w, h = 1920, 1080
    for n, spp in ((1024, 16), (8192, 4)):
        sph, orig, target = scenes.random_spheres(n)
        sph = sph.copy()
        cam = host.compute_camera(orig, target, w, h)
        rng = np.random.default_rng(1)
        with api.RtContext(w, h) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            for _ in range(3):
                ctx.reset()
                ctx.render_pass(spp, copy=False)
            base = ctx.stats()["last_kernel_ms"]
            for k in (0, 1, 8, 40, 200):
                who = np.sort(rng.choice(np.arange(2, n), k, replace=False)) if k else []
                for how in ("calls", "one_range"):
                    ts = []
                    for f in range(12):
                        t0 = time.perf_counter()
                        if k:
                            sph["p"][who] += np.float32([0.01, 0.0, 0.01])
                            if how == "calls":
                                for i in who:
                                    ctx.update_spheres(int(i), sph[int(i):int(i) + 1], ctx.stream)
                            else:
                                ctx.update_spheres(0, sph, ctx.stream)
                        ctx.reset_async(ctx.stream)
                        ctx.render_async(spp, ctx.stream)
                        ctx.throttle(0)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    print(json.dumps({"spheres": n, "spp": spp, "moved": k, "how": how if k else "none", "frame_wall_ms_median": round(float(np.median(ts[2:])), 3),
                                      "steady_kernel_ms": round(base, 3)}), flush=True)
                    if not k:
                        break


if __name__ == "__main__":
    main()
