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
    grep -h "gather stress\|WRONG\|OVERLAP\|un-reset\|Traceback\|Error" "$OUT/$name.log" | head -40 | tee -a "$OUT/summary.txt"
}
# 1. the failing chain, device wall-clock log only (no host-side logging that would change the timing)
run old_q${QUEUES}_timelog GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1
# 2. the same with the runtime's AQL packet log (barrier bit, acquire/release scopes, hardware queue per packet)
mkdir -p "$OUT/aql"
echo "=== aql log run" | tee -a "$OUT/summary.txt"
env GPU_MAX_HW_QUEUES=$QUEUES RT_OLD_RESET=kernel RT_PROBE=1 RT_TIMELOG=1 AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x801A \
    timeout -k 10 420 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
    --redirects 3 --log-dir "$OUT/aql" tools/gather_stress.py 240 6 gloo > "$OUT/aql_run.log" 2>&1
echo "exit $?" >> "$OUT/aql_run.log"
grep -h "gather stress\|WRONG\|OVERLAP\|un-reset" "$OUT/aql_run.log" $(find "$OUT/aql" -name "stdout.log") 2>/dev/null | head -40 | tee -a "$OUT/summary.txt"
for f in $(find "$OUT/aql" -name "stderr.log"); do
    grep -c "" "$f" >> "$OUT/summary.txt"
    gzip -9 "$f"
done
du -sh "$OUT" | tee -a "$OUT/summary.txt"
# 3. today's product path (rt_reset_async reads the pristine stream in place), same queue pressure, no env help
run new_q${QUEUES} GPU_MAX_HW_QUEUES=$QUEUES
run new_default
