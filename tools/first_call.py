#!/usr/bin/env python3
"""Where the first call's time goes: HIP start-up, the library's first context (allocations, kernel modules), the first
render.  python tools/first_call.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = [time.perf_counter()]
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(C.byref(p), 1 << 20)
t.append(time.perf_counter())                       # HIP runtime up
from raytracing_simple_amd import api, host  # noqa: E402

lib = api.load_library(diag=True)
t.append(time.perf_counter())                       # import + dlopen
ctx = api.RtContext(1920, 1080, diag=True)
t.append(time.perf_counter())                       # first context
phases = (C.c_double * 8)()
lib.rt_debug_create_breakdown(phases)
first_create = dict(zip(["device query", "stream + events", "device allocations", "kernel function attributes (code object load)",
                         "seed stream on the host", "seed upload", "restore kernel + wait", "total"], [round(v, 2) for v in phases]))
ctx.set_scene(host.demo_scene())
ctx.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 1920, 1080))
ctx.render_pass(64)
t.append(time.perf_counter())                       # first frame
ctx.reset()
ctx.render_pass(64)
t.append(time.perf_counter())
ctx2 = api.RtContext(1920, 1080, diag=True)
t.append(time.perf_counter())
lib.rt_debug_create_breakdown(phases)
second_create = [round(v, 2) for v in phases]
names = ["HIP start-up", "import + dlopen", "first context", "scene + first frame", "second frame", "second context"]
print({n: round((b - a) * 1e3, 1) for n, a, b in zip(names, t, t[1:])})
print({"first rt_create by phase (ms)": first_create, "second rt_create, same phases": second_create})
