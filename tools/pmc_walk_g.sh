#!/bin/bash
# Counters of the L2 walk (rt_trace_parity_pairs_g and its A/B arms): what a latency- or cache-bandwidth-bound kernel needs on top of
# tools/pmc_walk.sh's issue counters -- the vector L1 (TCP) request / hit / stall / latency counters, the address and data units, L2.
#   tools/pmc_walk_g.sh CONFIG MODES [TAG]      CONFIG: a tools/ab_bench.py configuration (r8192, r65536, r262144); MODES as in pmc_walk.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-r8192}; MODES=${2:-0}; TAG=${3:-pmc_walk_g}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $R/tools/pmc_modes.py $CFG $MODES > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
pass p1 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass p2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE
pass p3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum
pass p4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass p5 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
# (no TA_* / TD_* passes: rocprofv3 aborts on them on this image -- signal 6 -- and then sits until its time limit, 5 minutes per pass: gpurun_out/r06a)
pass p8 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass p9 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum
pass p10 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum SQ_VMEM_TA_ADDR_FIFO_FULL
python3 $R/tools/pmc_last.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
