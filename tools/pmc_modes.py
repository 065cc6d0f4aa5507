#!/usr/bin/env python3
"""Render config C (default c2) once per mode; run under rocprofv3 --pmc ... to compare instances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
cname = sys.argv[1]
modes = [int(m) if m.lstrip("-").isdigit() else api.instance_mode(m) for m in sys.argv[2].split(",")]
maker, w, h, spp = CONFIGS[cname]
sph, orig, target = maker()
cam = host.compute_camera(orig, target, w, h)
with api.RtContext(w, h, diag=any(m >= 100 for m in modes) or bool(os.environ.get('RT_ORDER'))) as ctx:
    ctx.set_scene(sph); ctx.set_camera(cam)
    warm = int(os.environ.get("RT_PMC_WARM", "3"))      # frames before the measured one: tile costs, the heavy-first order
    if os.environ.get("RT_ORDER"):                         # 0 / 1 (rt_debug_set_tile_order)
        ctx._check(ctx._lib.rt_debug_set_tile_order(ctx._h, int(os.environ["RT_ORDER"], 0)))
    for m in modes:
        ctx.set_mode(m)
        for _ in range(warm + 1):
            ctx.reset(); ctx.render_pass(spp, copy=False)
        for _ in range(int(os.environ.get("RT_PMC_UNSEEN", "0"))):      # launches on passes not rendered before (the LAST launches of the run)
            ctx.render_pass(spp, copy=False)
        st = ctx.stats()
        print("MODE", m, st, flush=True)
