#!/bin/bash
# LDS / VALU counters of the instance a configuration renders with: tools/pmc_bvh.sh CONFIG   (two counter passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}
OUT=$R/gpurun_out/pmc_bvh
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u > "$OUT/lds_counters.txt"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$OUT/p1" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/p2" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; }
python3 $R/tools/pmc_last.py "$OUT"
