#!/usr/bin/env python3
"""The plain sweep over a table beyond LDS (rt_trace_parity_g) on the nan9800 scene at several image sizes and pass counts: kernel time from the
context's events and what the same lane tests would cost the vector ALU at full width.  `python tools/g_sweep_probe.py [other librt_hip.so]` --
the optional argument renders with another build of the library (the A/B against the library before the change: profiles/r06_g_sweep_forms.jsonl)."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_simple_amd import api, host
import bench
if len(sys.argv) > 1:          # another build of the library (an A/B against a frozen one)
    _alt = os.path.abspath(sys.argv[1])
    api.lib_path = lambda diag=False: _alt
    print('library', _alt, flush=True)
sph, orig, target = bench.nan_scene(9800)
for (w,h,spp) in [(640,360,1),(640,360,4),(640,360,16),(1920,1080,1),(1920,1080,4)]:
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w,h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        ts=[]
        for k in range(4):
            ctx.reset()
            ctx.render_pass(spp, copy=False)
            st=ctx.stats()
            ts.append(st["last_kernel_ms"])
        tests=st["sphere_tests"]
        floor_ms = tests/64*16*2/(1024*2.1e9)*1e3
        print(json.dumps(dict(w=w,h=h,spp=spp,kernel=ctx.last_kernel,ms=[round(t,3) for t in ts],tests=tests,valu_floor_ms_at_full_lanes=round(floor_ms,3))), flush=True)
