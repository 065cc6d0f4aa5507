#!/bin/bash
# The round-1 reset->launch stale-seed failure, instrumented: which pair of executions
# overlaps in device time, and what the runtime put into the AQL packets.  One-GPU box, 4 ranks on device 0.
#   bash tools/stale_seed_experiment.sh <out_dir> [frames] [queues]
set -u
OUT=${1:-gpurun_out/stale}; FRAMES=${2:-480}; QUEUES=${3:-4}
mkdir -p "$OUT"
export RT_BENCH_SINGLE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # name, extra env...
    local name=$1; shift
    echo "=== $name: $*" | tee -a "$OUT/summary.txt"
    env "$@" timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
        tools/gather_stress.py "$FRAMES" 6 gloo > "$OUT/$name.log" 2>&1
    echo "exit $?" >> "$OUT/$name.log"
    grep -h "gather stress\|WRONG\|OVERLAP\|STALE\|timelog summary\|ran .* of 1024\|Traceback\|Error" "$OUT/$name.log" | cut -c1-1500 | head -60 | tee -a "$OUT/summary.txt"
}
# 1. the failing chain, device wall-clock log only (no host-side logging that would change the timing); twice
run old_q${QUEUES}_a GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1
run old_q${QUEUES}_b GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1
# 2. which side loses the data?  (a) the producer's waves write their XCD's L2 back themselves; (b) the consumer's
#    waves invalidate themselves; (c) both
run old_q${QUEUES}_copy_release_a GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_COPY_RELEASE=1
run old_q${QUEUES}_copy_release_b GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_COPY_RELEASE=1
run old_q${QUEUES}_probe_acquire_a GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_PROBE_ACQUIRE=1
run old_q${QUEUES}_probe_acquire_b GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_PROBE_ACQUIRE=1
#    (d) the producer stores write-through (sc1): nothing of it is ever a dirty L2 line
run old_q${QUEUES}_copy_wt_a GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_COPY_WT=1
run old_q${QUEUES}_copy_wt_b GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_COPY_WT=1
run old_q${QUEUES}_copy_wt_c GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 RT_COPY_WT=1
# 3. queue pressure: the default queue count and one queue per stream
run old_qdefault -u GPU_MAX_HW_QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1
run old_q24 GPU_MAX_HW_QUEUES=24 RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1
# 4. the runtime's AQL packet log (barrier bit, acquire/release scopes, hardware queue per packet) of one short run
mkdir -p "$OUT/aql"
echo "=== aql log run" | tee -a "$OUT/summary.txt"
env GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 AMD_LOG_LEVEL=4 AMD_LOG_MASK=32794 \
    timeout -k 10 420 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
    --redirects 3 --log-dir "$OUT/aql" tools/gather_stress.py 120 6 gloo > "$OUT/aql_run.log" 2>&1
echo "exit $?" >> "$OUT/aql_run.log"
grep -h "gather stress\|WRONG\|OVERLAP\|STALE\|timelog summary" "$OUT/aql_run.log" $(find "$OUT/aql" -name "stdout.log") 2>/dev/null | head -40 | tee -a "$OUT/summary.txt"
for f in $(find "$OUT/aql" -name "stderr.log"); do
    grep -c "" "$f" >> "$OUT/summary.txt"
    # keep the packet lines only (dispatch / barrier headers), drop addresses nobody will read
    grep "Header\|HWq\|queue" "$f" | cut -c1-400 | gzip -9 > "$f.packets.gz"
    rm -f "$f"
done
du -sh "$OUT" | tee -a "$OUT/summary.txt"
# 5. today's product path (rt_reset_async reads the pristine stream in place), same queue pressure, no env help
run new_q${QUEUES}_a GPU_MAX_HW_QUEUES=$QUEUES
run new_q${QUEUES}_b GPU_MAX_HW_QUEUES=$QUEUES
run new_q${QUEUES}_c GPU_MAX_HW_QUEUES=$QUEUES
run new_default -u GPU_MAX_HW_QUEUES
