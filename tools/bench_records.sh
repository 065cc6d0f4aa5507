#!/bin/bash
# The bench lines that go into profiles/<tag>_bench_*.json (run through gpurun AFTER tools/collect.py has written this library's
# counters into profiles/pmc_traffic.json, so that the lines carry `traffic` / `executed`):  tools/bench_records.sh TAG
set -u
TAG=${1:-bench}; O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rm -f $O/bench_workloads.jsonl
for wl in c16 c3 c4 c5 box120 r2048; do
    python bench.py --workload $wl --steps 6 --warmup 3 --no-cpu >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
d = json.load(open(O + "/bench_default.json")); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"], r.get("counters_withheld"), d["unseen_passes"]["kernel_ms"], d["frames_in_flight"]["ms_per_step"], d["frames_in_flight"]["value"])
print(d["north_star_target"]["ms_per_step"], d["north_star_target"]["value"], d["large_scene"]["ms_per_step"], d["large_scene"]["value"], d["large_scene"]["moving_scene"]["kernel_ms"], d["other_mode"]["ms_per_step"], d["other_mode"]["value"])
print({k: (v["ms"], v["ms_on_a_live_context"]) for k, v in d["first_frame"].items() if isinstance(v, dict)})
print(d["config"]["strong_scaling_bound"]["slowest_wavefront_ms"], d["config"]["strong_scaling_bound"]["predicted_speedup_ceiling"], d["cpu_baseline"]["ms_per_frame"], d["cpu_baseline"]["port"]["ms_per_frame"])
for l in open(O + "/bench_workloads.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"][:40], d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["traffic"], d.get("unseen_passes", {}).get("kernel_ms"))
PY
