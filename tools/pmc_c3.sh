#!/bin/bash
# PMC passes of one C3 frame (1024 spheres): is the sweep bound by the VALU or by LDS broadcast reads?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}
OUT=$R/gpurun_out/pmc_$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/$name.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU && \
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 && \
pass sq4 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT
python3 $R/tools/pmc_last.py "$OUT"
