#!/bin/bash
# L2 hit rate of the LDS-staged sphere reads IN ISOLATION (north_star): the render kernels' table staging alone
# (rt_debug_stage_tables: every workgroup of the launch grid reads the scene tables into LDS and does nothing else) under
# rocprofv3 --pmc, next to the whole render kernel's counters on the same scene.   tools/pmc_staging.sh [c2,c16,c5,...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFGS=${1:-c2,c16,c5}
OUT=$R/gpurun_out/pmc_staging
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/p" -- python3 $R/tools/staging_probe.py $CFGS > "$OUT/p.log" 2>&1 || { tail -5 "$OUT/p.log"; exit 1; }
grep STAGE "$OUT/p.log"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("rt_stage_probe", "rt_trace"))]
by = {}
for r in rows:
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Grid_Size"], r["LDS_Block_Size"]), {}).setdefault(r["Counter_Name"], 0.0)
    by[(int(r["Dispatch_Id"]), r["Kernel_Name"], r["Grid_Size"], r["LDS_Block_Size"])][r["Counter_Name"]] += float(r["Counter_Value"])
order, stage_rates = [], {}
for line in open(os.path.join(sys.argv[1], "p.log")):
    if line.startswith("STAGE"):
        order.append(line.split()[1])
probes = [k for k in sorted(by) if k[1].startswith("rt_stage_probe")]
for i, k in enumerate(probes):                 # three staging launches per configuration, in the order of the STAGE lines
    d = by[k]
    if i // 3 < len(order):
        h, m = stage_rates.setdefault(order[i // 3], [0.0, 0.0])
        stage_rates[order[i // 3]] = [h + d.get("TCC_HIT_sum", 0.0), m + d.get("TCC_MISS_sum", 0.0)]
if os.environ.get("RT_STAGING_JSON"):
    import json
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
    from raytracing_simple_amd import api
    json.dump({"build_id": api.build_id(diag=True), "what": "TCC hit rate of the render kernels' table staging alone (rt_debug_stage_tables, three launches of the "
               "render grid per configuration, rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum): north_star's 'L2-hit rate on the LDS-staged sphere reads'",
               "l2_hit_rate_staged_tables": {k: round(h / max(h + m, 1.0), 4) for k, (h, m) in stage_rates.items()}},
              open(os.environ["RT_STAGING_JSON"], "w"), indent=1)
for k in sorted(by):
    d = by[k]
    hit, miss = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
    print(f"dispatch {k[0]:4d} {k[1]:28s} grid {k[2]:>9s} lds {k[3]:>6s}  TCC_REQ {d.get('TCC_REQ_sum', 0):12.0f}  HIT {hit:12.0f}  MISS {miss:10.0f}  hit rate {hit / max(hit + miss, 1):.4f}  EA read requests {d.get('TCC_EA0_RDREQ_sum', 0):10.0f}")
PY
