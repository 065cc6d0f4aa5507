#!/bin/bash
# Round-2 measurement session on the GPU box (run through gpurun): part A = suite, bench lines, A/B table;
# part B = rocprofv3 trace + PMC passes, native hosts, 2-rank rehearsal.  Outputs under gpurun_out/<tag>/.
set -u
PART=${1:-A}; O=gpurun_out/${2:-r02final}; mkdir -p $O
if [ "$PART" = "A" ]; then
    python -m pytest tests -m gpu -q --timeout 900 --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
    tail -3 $O/pytest.log
    [ $rc -le 1 ] || exit $rc
    python bench.py --steps 20 --warmup 5 > $O/bench_parity.json 2> $O/bench_parity.err && tail -c 600 $O/bench_parity.json
    python bench.py --steps 20 --warmup 5 --mode fast --no-cpu > $O/bench_fast.json 2> $O/bench_fast.err
    for wl in c16 c3 c4 c5; do
        python bench.py --workload $wl --steps 6 --warmup 2 >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err
    done
    python tools/ab_bench.py --configs c2,c16,c64,c256,c3,c5,c4 --modes 0,1 --orders 1 --rounds 5 > $O/ab_all_configs.jsonl 2>&1
    cut -c1-200 $O/ab_all_configs.jsonl
else
    bash tools/profile_gpu.sh ${2:-r02final}/prof_parity parity > $O/prof_parity.log 2>&1; tail -2 $O/prof_parity.log
    bash tools/profile_gpu.sh ${2:-r02final}/prof_fast fast > $O/prof_fast.log 2>&1; tail -2 $O/prof_fast.log
    raytracing_simple_amd/rt_bench 2 1 0 --w 1920 --h 1080 --spp 64 --oneshot 8 | tee $O/rt_bench_oneshot.json
    raytracing_simple_amd/rt_bench 2 1 0 raytracing_simple_amd/scenes_scn/c16_demo_plus_10.scn --no-doubling --w 1920 --h 1080 --spp 64 | tee $O/rt_bench_c16.json
    raytracing_simple_amd/rt_inflight 1 20 | tee $O/rt_inflight.jsonl; raytracing_simple_amd/rt_inflight 2 20 | tee -a $O/rt_inflight.jsonl
    python tools/ab_bench.py --configs c2,c16 --modes 0 --gates 8,10,12,14 --orders 1 --rounds 9 > $O/ab_gates.jsonl 2>&1; cut -c1-200 $O/ab_gates.jsonl
    RT_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err; tail -c 400 $O/bench_n2_rehearsal.json
fi
echo "part $PART done"
