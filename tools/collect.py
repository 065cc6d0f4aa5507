#!/usr/bin/env python3
"""After `gpurun -- bash tools/session.sh P TAG`: summarise gpurun_out/TAG/prof_* into profiles/<prefix>_* and profiles/pmc_traffic.json
(what bench.py reads for `roofline.traffic` / `roofline.executed`, each record stamped with the build id of the library it was
measured on).   python tools/collect.py TAG PREFIX      e.g.  python tools/collect.py r04p r04p"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
KERNEL = {("c2", "parity"): "rt_trace_parity_w1", ("c2", "fast"): "rt_trace_fast_w1", ("c16", "parity"): "rt_trace_parity_coop_w1",
          ("c3", "parity"): "rt_trace_parity_pairs", ("c5", "parity"): "rt_trace_parity_coop_w1"}
LABEL = {"c2": "C2: Demo, 1920x1080, 64 spp", "c16": "north-star target: 16 spheres, 1920x1080, 64 spp", "c3": "C3: 1024 spheres, 1920x1080, 16 spp",
         "c5": "C5: 64-sphere mirror box, 1920x1080, 64 spp"}
for (wl, mode), kern in KERNEL.items():
    src = os.path.join(tag, f"prof_{wl}_{mode}")
    if not os.path.isdir(os.path.join(ROOT, "gpurun_out", src)):
        print("missing", src)
        continue
    env = dict(os.environ, RT_PMC_KEY=wl)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), src, mode, f"{prefix}_{wl}_{mode}", kern, LABEL[wl]],
                   env=env, check=True, stdout=subprocess.DEVNULL)
    print("profiles/%s_%s_%s.{md,json}" % (prefix, wl, mode))
