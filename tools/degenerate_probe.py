#!/usr/bin/env python3
"""Non-finite and degenerate inputs through the HIP path and the oracle (information, not a gate)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
from raytracing_simple_amd import api, host
w, h, spp = 48, 32, 3
def run(tag, sph, orig, target):
    cam = host.compute_camera(orig, target, w, h)
    with np.errstate(all="ignore"):
        want = O.render(sph, cam, w, h, spp)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        px = ctx.render_pass(spp); col = ctx.read_colors(); sd = ctx.read_seeds()
    a, b = col.view(np.uint32), want["colors"].view(np.uint32)
    nan_a, nan_b = np.isnan(col), np.isnan(want["colors"])
    print(f"{tag:34s} pixels {'same' if np.array_equal(px, want['pixels']) else 'DIFF'}  seeds {'same' if np.array_equal(sd, want['seeds']) else 'DIFF'}  "
          f"colours bitwise {'same' if np.array_equal(a, b) else 'DIFF'}  NaN positions {'same' if np.array_equal(nan_a, nan_b) else 'DIFF'} ({int(nan_b.sum())} NaN)  "
          f"non-NaN colours {'same' if np.array_equal(a[~nan_b], b[~nan_b]) else 'DIFF'}", flush=True)
demo = host.demo_scene()
run("camera orig == target", demo, (1.0, 2.0, 3.0), (1.0, 2.0, 3.0))
s = demo.copy(); s["rad"][1] = np.nan
run("one radius NaN", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["p"][2] = (1e38, 1e38, 1e38)
run("one centre at 1e38 (overflow)", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["rad"][0] = np.inf
run("ground radius inf", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["c"][0] = (np.nan, 0.5, 0.5)
run("NaN colour", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["e"][5] = (1e38, 1e38, 1e38)
run("emission 1e38 (inf radiance)", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["rad"][3] = -10.0
run("negative radius", s, host.DEMO_ORIG, host.DEMO_TARGET)
s = demo.copy(); s["rad"][:] = 1e-30
run("subnormal-scale radii", s, host.DEMO_ORIG, host.DEMO_TARGET)
run("camera at 1e30", demo, (1e30, 1e30, 1e30), (0.0, 0.0, 0.0))
