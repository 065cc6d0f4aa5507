#!/bin/bash
# A round's measurement session on the GPU box (run through gpurun).  Outputs under gpurun_out/<tag>/; tools/collect.py (run
# afterwards, in the repository) turns part P into profiles/<round>p_* and profiles/pmc_traffic.json.
#   A   suite, bench lines (default, fast, the other workloads), in-process A/B table, native hosts, 2-rank rehearsal
#   P   rocprofv3 --kernel-trace --stats + separate --pmc passes of bench.py for every kernel the bench line names:
#       C2 (parity, fast), the 16-sphere target, C3, C5
#   M   the in-library multi-device rehearsal under --kernel-trace (the gather of frame k against the render of frame k + 1)
set -u
PART=${1:-A}; TAG=${2:-session}; O=gpurun_out/$TAG; mkdir -p $O
if [ "$PART" = "A" ]; then
    python -m pytest tests -m gpu -q --timeout 900 --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
    tail -3 $O/pytest.log
    [ $rc -le 1 ] || exit $rc
    python bench.py --steps 20 --warmup 5 > $O/bench_parity.json 2> $O/bench_parity.err && tail -c 400 $O/bench_parity.json
    python bench.py --steps 20 --warmup 5 --mode fast --no-cpu > $O/bench_fast.json 2> $O/bench_fast.err
    for wl in c16 c3 c4 c5; do
        python bench.py --workload $wl --steps 6 --warmup 3 >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err
    done
    python tools/ab_bench.py --configs c2,c16,c64,c256,c3,c5,c4 --modes 0,1 --rounds 5 > $O/ab_all_configs.jsonl 2>&1
    cut -c1-220 $O/ab_all_configs.jsonl
    raytracing_simple_amd/rt_bench 2 1 0 --w 1920 --h 1080 --spp 64 --oneshot 8 | tee $O/rt_bench_oneshot.json
    raytracing_simple_amd/rt_inflight 1 20 | tee $O/rt_inflight.jsonl; raytracing_simple_amd/rt_inflight 2 20 | tee -a $O/rt_inflight.jsonl
    RT_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err; tail -c 300 $O/bench_n2_rehearsal.json
elif [ "$PART" = "P" ]; then
    # every shipped parity instance has a stamped record: _w1 (c2), _coop_w1 (c16, c5), _pairs (c3), _coop (box120), _pairs_m (r2048), _pairs_g (r8192),
    # rt_trace_parity_g (nan9800) -- tests/test_abi.py holds the set to the library's build id
    # (RT_PROF_SPECS="c2 parity;c3 parity" limits the call to some of them)
    if [ -z "${RT_PROF_SPECS:-}" ] || [ "${RT_PROF_STAGING:-0}" = "1" ]; then
        RT_STAGING_JSON=$(pwd)/$O/pmc_staging.json bash tools/pmc_staging.sh c2,c16,c5,box120 > $O/pmc_staging.log 2>&1; tail -4 $O/pmc_staging.log
    fi
    IFS=';' read -ra SPECS <<< "${RT_PROF_SPECS:-c2 parity;c2 fast;c16 parity;c3 parity;c5 parity;box120 parity;r2048 parity;r8192 parity;nan9800 parity;nan9800hd parity;dust10k parity}"
    for spec in "${SPECS[@]}"; do
        set -- $spec
        bash tools/profile_gpu.sh $TAG/prof_$1_$2 $2 $1 > $O/prof_$1_$2.log 2>&1; tail -1 $O/prof_$1_$2.log
    done
else
    export TMPDIR=/tmp; R=$(pwd); cd /tmp
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/multi_trace -- python3 $R/tools/multi_overlap.py > $R/$O/multi_overlap.log 2>&1
    cd $R; python3 tools/multi_overlap.py --summarise $O/multi_trace | tee $O/multi_overlap.json
fi
echo "part $PART done"
