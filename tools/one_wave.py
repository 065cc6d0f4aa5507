#!/usr/bin/env python3
"""Render a tiny image (few wavefronts) so PMC counts can be read per wavefront.
usage: one_wave.py W H SPP MODE [config]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
w, h, spp, mode = (int(v) for v in sys.argv[1:5])
maker = CONFIGS[sys.argv[5] if len(sys.argv) > 5 else "c2"][0]
sph, orig, target = maker()
cam = host.compute_camera(orig, target, w, h)
with api.RtContext(w, h, diag=True) as ctx:
    ctx.set_scene(sph); ctx.set_camera(cam); ctx.set_mode(mode)
    ctx.render_pass(spp, copy=False)
    st = ctx.stats()
    buf = (C.c_ulonglong * 24)()
    api.load_library(diag=True).rt_debug_counters(ctx._h, buf)
    print("stats", st)
    print("census", [(c >> 32, (c & 0xFFFFFFFF)) for c in list(buf)[:10]])
