#!/usr/bin/env python3
"""F contexts in flight on their own streams (bench.py's pattern), every frame checked."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_simple_amd import api, host, scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w, h, spp = 480, 270, 16
scn = [scenes.demo_plus(16), (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET)]
ctxs, want = [], []
for k in range(F):
    sph, orig, target = scn[k % 2]
    cam = host.compute_camera(orig, target, w, h)
    c = api.RtContext(w, h, diag=True); c.set_scene(sph); c.set_camera(cam); ctxs.append(c)
    with api.RtContext(w, h, diag=True) as ref:
        ref.set_scene(sph); ref.set_camera(cam); want.append(ref.render_pass(spp))
bad = 0
progressive = len(sys.argv) > 3 and sys.argv[3] == "progressive"
for r in range(rounds):
    if progressive:          # launches that depend on each other through seeds and colours, queued without waiting
        for c in ctxs:
            c.reset_async(c.stream)
        for _ in range(spp // 2):
            for c in ctxs:
                c.render_async(2, c.stream)
        torch.cuda.synchronize()
        for k, c in enumerate(ctxs):
            if not np.array_equal(c.render_pass(0), want[k]):
                bad += 1; print("round", r, "context", k, "differs (progressive)", flush=True)
        continue
    for c in ctxs:
        if len(sys.argv) > 3 and sys.argv[3] in ("copykernel", "memcpy"):    # the earlier reset: seeds restored by a copy, read back by the launch
            import ctypes as C
            lib = api.load_library(diag=True)
            
            lib.rt_debug_reset_by_copy(c._h, C.c_void_p(c.stream), 1 if sys.argv[3] == "memcpy" else 0)
        else:
            c.reset_async(c.stream)
        c.render_async(spp, c.stream)
    torch.cuda.synchronize()
    for k, c in enumerate(ctxs):
        if not np.array_equal(c.render_pass(0), want[k]):
            bad += 1; print("round", r, "context", k, "differs", flush=True)
print("in-flight stress:", F, "contexts x", rounds, "rounds,", bad, "problems")
