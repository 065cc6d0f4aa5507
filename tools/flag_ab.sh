#!/bin/bash
# On the GPU box: time every library tools/flag_ab.py built (the box's copy of the tree is scratch: each variant is copied over
# the diagnostics library in turn).  tools/flag_ab.sh "c2,c16,c3" > gpurun_out/<tag>/flag_ab.jsonl
set -u
CFG=${1:-c2,c16,c3}
D=raytracing_simple_amd/csrc/_obj/flagab
cp raytracing_simple_amd/librt_hip_diag.so /tmp/librt_hip_diag.keep
for v in $(ls $D); do
    [ -f $D/$v/librt_hip_diag.so ] || continue
    cp $D/$v/librt_hip_diag.so raytracing_simple_amd/librt_hip_diag.so
    python3 tools/ab_bench.py --configs $CFG --modes 0 --rounds ${RT_AB_ROUNDS:-5} 2>&1 | sed "s/^{/{\"flags\": \"$v\", /"
done
cp /tmp/librt_hip_diag.keep raytracing_simple_amd/librt_hip_diag.so
