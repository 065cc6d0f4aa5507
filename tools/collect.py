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
KERNEL.update({("box120", "parity"): "rt_trace_parity_coop", ("r2048", "parity"): "rt_trace_parity_pairs_m", ("r8192", "parity"): "rt_trace_parity_pairs_g",
               ("nan9800", "parity"): "rt_trace_parity_g", ("nan9800hd", "parity"): "rt_trace_parity_g", ("dust10k", "parity"): "rt_trace_parity_pairs_g"})
LABEL.update({"box120": "closed box of 120 mirror / glass spheres, 1920x1080, 8 spp", "r2048": "2048 random spheres, 1920x1080, 8 spp",
              "r8192": "8192 random spheres, 1920x1080, 4 spp (hierarchy read from HBM / L2)",
              "nan9800": "Demo scene + 9794 records with a NaN centre, 640x360, 1 spp (no hierarchy, table beyond LDS: the plain sweep over HBM / L2)",
              "nan9800hd": "Demo scene + 9794 records with a NaN centre, 1920x1080, 1 spp (the plain sweep over a table beyond LDS on a filled GPU)",
              "dust10k": "6000 small spheres among 4000 objects fifty times their size, 1920x1080, 4 spp (two size classes, both in the hierarchy)"})


def kernel_of(src, default):
    """The kernel the profiled bench.py run itself named (its JSON line ends trace.log): the library picks the instance."""
    import json
    try:
        lines = [l for l in open(os.path.join(ROOT, "gpurun_out", src, "trace.log")) if l.startswith("{")]
        return json.loads(lines[-1])["roofline"]["kernel"]
    except (OSError, ValueError, KeyError, IndexError):
        return default


for (wl, mode), kern in KERNEL.items():
    src = os.path.join(tag, f"prof_{wl}_{mode}")
    if not os.path.isdir(os.path.join(ROOT, "gpurun_out", src)):
        print("missing", src)
        continue
    kern = kernel_of(src, kern)
    env = dict(os.environ, RT_PMC_KEY=wl)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), src, mode, f"{prefix}_{wl}_{mode}", kern, LABEL[wl]],
                   env=env, check=True, stdout=subprocess.DEVNULL)
    print("profiles/%s_%s_%s.{md,json}  (%s)" % (prefix, wl, mode, kern))
# the staging reads' L2 hit rate in isolation (tools/pmc_staging.sh of the same session), into the same records
stage = os.path.join(ROOT, "gpurun_out", tag, "pmc_staging.json")
if os.path.exists(stage):
    import json
    st = json.load(open(stage))
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(path))
    for wl, rate in st["l2_hit_rate_staged_tables"].items():
        rec = (data.get(wl) or {}).get("parity")
        if rec and rec.get("build_id") == st["build_id"]:
            rec["l2_hit_rate_staged_tables"] = rate
    json.dump(data, open(path, "w"), indent=1)
    json.dump(st, open(os.path.join(ROOT, "profiles", prefix + "_staging_l2_hit_rate.json"), "w"), indent=1)
    print("profiles/%s_staging_l2_hit_rate.json" % prefix)
