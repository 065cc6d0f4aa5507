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
for k in sorted(by):
    d = by[k]
    hit, miss = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
    print(f"dispatch {k[0]:4d} {k[1]:28s} grid {k[2]:>9s} lds {k[3]:>6s}  TCC_REQ {d.get('TCC_REQ_sum', 0):12.0f}  HIT {hit:12.0f}  MISS {miss:10.0f}  hit rate {hit / max(hit + miss, 1):.4f}  EA read requests {d.get('TCC_EA0_RDREQ_sum', 0):10.0f}")
PY
