#!/bin/bash
# VALU instruction counts of kernel instances on one config: tools/pmc_ab.sh CONFIG MODES   (MODES: 0, 1, or kernel symbols of the diagnostics library)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}; MODES=${2:-0}
OUT=$R/gpurun_out/pmc_ab
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p" -- python3 $R/tools/pmc_modes.py $CFG $MODES > "$OUT/p.log" 2>&1 || { tail -5 "$OUT/p.log"; exit 1; }
python3 $R/tools/pmc_last.py "$OUT"
