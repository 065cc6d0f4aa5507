#!/bin/bash
# On the GPU box: the diagnostics library as built (leaves of 8) and the one tools/leaf_size_ab.py built (leaves of 4), same configurations
set -u
CFG=${1:-c3,c256}
MODES=${2:-rt_trace_parity_pairs}
cp raytracing_simple_amd/librt_hip_diag.so /tmp/librt_hip_diag.keep
for v in 8 4 8 4; do
    [ $v = 4 ] && cp raytracing_simple_amd/csrc/_obj/leaf4/librt_hip_diag.so raytracing_simple_amd/librt_hip_diag.so || cp /tmp/librt_hip_diag.keep raytracing_simple_amd/librt_hip_diag.so
    python3 tools/ab_bench.py --configs $CFG --modes $MODES --rounds ${RT_AB_ROUNDS:-4} 2>&1 | sed "s/^{/{\"leaf\": $v, /"
done
cp /tmp/librt_hip_diag.keep raytracing_simple_amd/librt_hip_diag.so
