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
with api.RtContext(w, h, diag=any(m >= 100 for m in modes)) as ctx:
    ctx.set_scene(sph); ctx.set_camera(cam)
    for m in modes:
        ctx.set_mode(m); ctx.reset(); ctx.render_pass(spp, copy=False)
        st = ctx.stats()
        print("MODE", m, st, flush=True)
