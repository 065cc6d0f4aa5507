#!/usr/bin/env python3
"""Section shares of the render loop from the census instances of the diagnostics library:

    python tools/stamp_profile.py CONFIGS [KERNEL]     KERNEL = rt_trace_parity_census (default; scenes below 12 spheres),
                                                       rt_trace_parity_coop_census (12 and more: the 16-sphere scene, C5)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402

NAMES = ["camera ray", "closest sweep", "hit point/normal", "light sample", "shadow sweep", "light contrib",
         "diffuse bounce", "spec/refr", "loop trip", "accumulate", "closest roots", "shadow roots"]

args = [a for a in sys.argv[1:] if not a.startswith("--")]
for cname in (args[0] if args else "c2").split(","):
    maker, w, h, spp = CONFIGS[cname]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        for _ in range(3):                  # the shipped instance first: tile costs, the heavy-first order
            ctx.reset()
            ctx.render_pass(spp, copy=False)
        ctx.reset()
        ctx.set_mode(api.instance_mode(args[1] if len(args) > 1 else ("rt_trace_parity_coop_census" if len(sph) >= 12 else "rt_trace_parity_census")))
        ctx.render_pass(spp, copy=False)
        st = ctx.stats()
        buf = (C.c_ulonglong * 24)()
        api.load_library(diag=True).rt_debug_counters(ctx._h, buf)
        v = list(buf)[:12]
        print(f"{cname}: {st['last_kernel_ms']:.3f} ms (census build); per section: wave-level executions, "
              f"active lanes per execution, executions per sample-wave")
        waves_samples = st["samples"] / 64.0
        for n, c in zip(NAMES, v):
            execs, lanes = c >> 32, c & 0xFFFFFFFF
            if n.endswith("roots"):
                sweeps = (v[1] if n.startswith("closest") else v[4]) >> 32
                print(f"  {n:18s} root halves {c:12d}  per sweep execution {c / max(sweeps, 1):6.2f} of {len(sph)} spheres")
            elif execs:
                print(f"  {n:18s} execs {execs:12d}  lanes/exec {lanes / execs:6.2f}  execs per 64 samples {execs / waves_samples:6.3f}")
