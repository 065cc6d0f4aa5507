#!/bin/bash
# Issued VALU instructions and active lanes of the LAST launch of a configuration's kernel: the same frame rendered again, and a
# launch on passes not rendered before (tools/pmc_modes.py RT_PMC_UNSEEN), deal of pixels off / as shipped:  tools/pmc_unseen.sh CONFIG
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}
export TMPDIR=/tmp
cd /tmp
for deal in 0 default; do for unseen in 0 2; do
    OUT=$R/gpurun_out/pmc_unseen_${CFG}_${deal}_${unseen}; rm -rf "$OUT"; mkdir -p "$OUT"
    if [ $deal = 0 ]; then export RT_NO_DEAL=1; else unset RT_NO_DEAL; fi
    RT_PMC_UNSEEN=$unseen timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/p" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/p.log" 2>&1 || { tail -3 "$OUT/p.log"; exit 1; }
    echo "== $CFG deal=$deal launches_on_unseen_passes=$unseen"
    python3 $R/tools/pmc_last.py "$OUT" | grep -v _coop
done; done
