#!/bin/bash
# Round 6's measurement records with ONE library (run through gpurun after the last change to csrc/): everything DESIGN.md quotes that is not a
# rocprofv3 profile (tools/session.sh P makes those).  Output: gpurun_out/<tag>/*.jsonl|txt, copied into profiles/r06_* by hand afterwards.
set -u
TAG=${1:-r06r}; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 "$@" > $O/$name 2> $O/$name.err || { echo "$name failed ($?)"; tail -3 $O/$name.err; }; }
run shard_prediction.jsonl python tools/shard_prediction.py --workloads c2,c4
run reference_scenes.jsonl python tools/reference_scenes.py
run fast_gate.jsonl python tools/fast_gate.py
: > $O/l2_walk_arms.jsonl
for cfg in r8192 r65536 r262144; do
    timeout -k 10 300 python tools/walk_ab.py $cfg base inst=rt_trace_parity_pairs_gt inst=rt_trace_parity_pairs_gp inst=rt_trace_parity_pairs_gtp inst=rt_trace_parity_pairs_gq >> $O/l2_walk_arms.jsonl 2>> $O/l2_walk_arms.err
done
timeout -k 10 300 python tools/walk_ab.py r3000 base l2 l2,inst=rt_trace_parity_pairs_gq l2,inst=rt_trace_parity_pairs_gt >> $O/l2_walk_arms.jsonl 2>> $O/l2_walk_arms.err
for top in 63 127; do
    RT_TOP_PAIRS=$top timeout -k 10 300 python tools/walk_ab.py r8192 base inst=rt_trace_parity_pairs_gt | sed "s/\"arm\": \"inst=rt_trace_parity_pairs_gt\"/\"arm\": \"inst=rt_trace_parity_pairs_gt, RT_TOP_PAIRS=$top\"/" >> $O/l2_walk_arms.jsonl 2>> $O/l2_walk_arms.err
done
echo "== l2 walk arms done"
: > $O/l2_walk_knobs.jsonl
for cfg in r8192 r65536; do
    timeout -k 10 300 python tools/walk_ab.py $cfg base tail=2 tail=4 tail=8 tail=16 gate=8 gate=32 round=2 round=8 >> $O/l2_walk_knobs.jsonl 2>> $O/l2_walk_knobs.err
done
timeout -k 10 300 python tools/walk_ab.py r262144 base tail=4 tail=16 gate=32 round=8 >> $O/l2_walk_knobs.jsonl 2>> $O/l2_walk_knobs.err
run order_short_launches.jsonl python tools/order_short_launches.py
python tools/order_short_launches.py --long-first >> $O/order_short_launches.jsonl 2>> $O/order_short_launches.jsonl.err
run lds_budget_ab.jsonl python tools/lds_budget_ab.py --sizes 1200,1600,2048,2400,3000 --scn complex
: > $O/coop_threshold_ab.jsonl
for cm in 0 1; do
    timeout -k 10 300 python tools/ab_bench.py --configs c2,c9,c10,c11,c12 --modes 0 --coop-min $cm --rounds 3 | sed "s/^{/{\"coop_min\": $cm, \"what\": \"cooperative any-hit $( [ $cm = 0 ] && echo off || echo on ) whatever the sphere count (rt_debug_set_coop_min)\", /" >> $O/coop_threshold_ab.jsonl 2>> $O/coop_threshold_ab.err
done
# the measurement of coop against plain inside a scene's first long frame (what launch_small does NOT do) against the deferred one
run coop_probe_first_frame.jsonl python tools/first_frame_probe.py
bash tools/pmc_walk_g.sh r8192 rt_trace_parity_pairs_g,rt_trace_parity_pairs_gq,rt_trace_parity_pairs_gt $TAG/pmc_l2 > $O/l2_walk_pmc.txt 2>&1
run bench_default.json python bench.py --steps 20 --warmup 5
for wl in c16 c3 c4 c5 r8192; do
    timeout -k 10 400 python bench.py --workload $wl --steps 6 --warmup 3 >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err
done
RT_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err
raytracing_simple_amd/rt_bench 2 1 0 --w 1920 --h 1080 --spp 64 --oneshot 9 > $O/rt_bench_oneshot.json 2>&1
echo "records $TAG done"
