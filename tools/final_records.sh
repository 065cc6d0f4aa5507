#!/bin/bash
# The bench lines of a round's final library (after tools/collect.py has stamped its counters): tools/final_records.sh TAG  ->  gpurun_out/TAG/
# (default line, the other BASELINE configurations and the instances they do not reach, the two-rank rehearsal on one device, rt_render one-shot)
set -u
TAG=${1:-final}; O=gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
: > $O/bench_other_workloads.jsonl
for wl in c16 c3 c4 c5 r8192 nan9800; do
    timeout -k 10 400 python bench.py --workload $wl --steps 6 --warmup 3 >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err; echo "$wl done"
done
timeout -k 10 300 python bench.py --workload nan9800hd --steps 6 --warmup 3 --no-cpu >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err
timeout -k 10 300 python bench.py --workload dust10k --steps 6 --warmup 3 --no-cpu >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err
RT_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err
raytracing_simple_amd/rt_bench 2 1 0 --w 1920 --h 1080 --spp 64 --oneshot 9 > $O/rt_bench_oneshot.json 2>&1
echo "records $TAG done"
