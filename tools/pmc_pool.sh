#!/bin/bash
# wave-state counters of the hierarchy walk at two pool heights: tools/pmc_pool.sh [CONFIG] [ROWS_A] [ROWS_B]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}; A=${2:-8}; B=${3:-64}
OUT=$R/gpurun_out/pmc_pool
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for rows in $A $B; do
  export RT_POOL_ROWS=$rows
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT/a_$rows" -- python3 $R/tools/pmc_modes.py $CFG rt_trace_parity_pairs > "$OUT/a_$rows.log" 2>&1 || { tail -5 "$OUT/a_$rows.log"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/b_$rows" -- python3 $R/tools/pmc_modes.py $CFG rt_trace_parity_pairs > "$OUT/b_$rows.log" 2>&1 || { tail -5 "$OUT/b_$rows.log"; }
  grep MODE "$OUT/a_$rows.log" | cut -c1-200
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "[ab]_*"))):
    if not os.path.isdir(d): continue
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("rt_trace"):
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(os.path.basename(d), " ".join(f"{k}={v:.4g}" for k, v in sorted(acc.items())))
PY
