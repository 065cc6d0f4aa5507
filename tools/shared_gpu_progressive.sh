#!/bin/bash
# Several processes sharing one GPU, each queueing its passes asynchronously on one stream (the
# adapter's display cadence): does every process end with the frame of a single fused launch?
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/raytracing_simple_amd/rt_bench
T=/tmp/rt_shared; mkdir -p $T
NP=${1:-4}; SPP=${2:-400}
$B 2 1 0 --w 640 --h 360 --spp $SPP --out $T/ref.ppm > /dev/null
bad=0
for round in 1 2 3 4 5 6; do
  for p in $(seq 1 $NP); do
    $B 2 1 0 --w 640 --h 360 --spp $SPP --passes-per-launch 1 --pin --readback-ms 8 --out $T/p$p.ppm > /dev/null &
  done
  wait
  for p in $(seq 1 $NP); do cmp -s $T/p$p.ppm $T/ref.ppm || { bad=$((bad+1)); echo "round $round process $p differs"; }; done
done
echo "shared-GPU progressive: $NP processes x 6 rounds x $SPP passes, $bad wrong final frames"
