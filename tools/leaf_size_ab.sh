#!/bin/bash
# Leaves of N spheres against leaves of 8: two builds of the libraries in turn (RT_BUILD_DEFINES), the same frames timed with each.
#   tools/leaf_size_ab.sh TAG "6 8"
set -u
TAG=${1:-leaf}; SIZES=${2:-"6 8"}; O=gpurun_out/$TAG; mkdir -p $O
for n in $SIZES 8; do
    RT_BUILD_DEFINES="-DRT_BVH_LEAF=$n" python -m raytracing_simple_amd._build --force > $O/build_$n.log 2>&1 || { tail -5 $O/build_$n.log; exit 1; }
    export RT_BUILD_DEFINES="-DRT_BVH_LEAF=$n"
    python -m pytest tests/test_gpu_bvh.py -m gpu -q -x -p no:cacheprovider -k "the_walk_equals_the_oracle or adversarial_scenes" 2>&1 | tail -1
    python tools/ab_bench.py --configs c3,c256,c64 --modes 0 --rounds 4 --unseen 2>&1 | sed "s/^{/{\"leaf\": $n, /" | cut -c1-200 | tee -a $O/leaf_size_ab.jsonl
    unset RT_BUILD_DEFINES
done
python -m raytracing_simple_amd._build --force > $O/build_default.log 2>&1
