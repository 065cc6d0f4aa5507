#!/usr/bin/env python3
"""For tools/pmc_staging.sh: per configuration one warm render, then the table staging alone (3 launches of the render kernel's
grid), then one render again -- the profiler lists the dispatches in this order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
for name in (sys.argv[1] if len(sys.argv) > 1 else "c2").split(","):
    maker, w, h, spp = CONFIGS[name]
    sph, orig, target = maker()
    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(host.compute_camera(orig, target, w, h))
        ctx.render_pass(spp, copy=False)
        ctx._check(ctx._lib.rt_debug_stage_tables(ctx._h, spp, 3))
        ctx.reset(); ctx.render_pass(spp, copy=False)
        print("STAGE", name, len(sph), "spheres;", ctx.last_kernel, flush=True)
