#!/bin/bash
# tools/profile_gpu.sh TAG MODE [WORKLOAD] -- run on the GPU box (via gpurun): kernel-trace stats + separate
# PMC passes of `bench.py --mode MODE --workload WORKLOAD` (default c2).  Writes CSVs under gpurun_out/TAG/.
# PMC passes never combine with anything but --kernel-trace (pool rule).
set -u
TAG=${1:-prof}
MODE=${2:-parity}
WL=${3:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# which library these counters describe (raytracing_simple_amd/_build.py source_hash; bench.py compares it with the library it runs)
python3 -c "import sys; sys.path.insert(0, '$R'); from raytracing_simple_amd import api; print(api.build_id())" > "$OUT/build_id.txt"
BENCH="python3 $R/bench.py --mode $MODE --workload $WL --steps 10 --warmup 3 --no-cpu --no-extras"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
  echo "pass $name ok"
}
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU && \
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 && \
pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT && \
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "profile $TAG done"
