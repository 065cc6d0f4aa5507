#!/bin/bash
# Counters of kernel instances on one configuration, three passes (instruction counts and lanes; waits; LDS):
#   tools/pmc_walk.sh CONFIG MODES [TAG]     MODES: 0, 1, or kernel symbols of the diagnostics library, comma-separated
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}; MODES=${2:-rt_trace_parity_pairs}; TAG=${3:-pmc_walk}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $R/tools/pmc_modes.py $CFG $MODES > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
pass p1 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY && \
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM && \
pass p3 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE
python3 $R/tools/pmc_last.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
