#!/bin/bash
# Run the steps of a GPU session one after the other: `tools/gpu_steps.sh TAG "seconds|name|command" ...`.  A step that fails with an ordinary
# error is reported and the next one runs; a step that is KILLED by its time limit stops the session (no further GPU step after a hang).
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for spec in "$@"; do
    IFS='|' read -r secs name cmd <<< "$spec"
    echo "== $name (limit ${secs}s)"; t0=$(date +%s)
    timeout -k 10 $secs bash -c "$cmd" > $O/$name.out 2> $O/$name.err; rc=$?
    echo "== $name rc=$rc $(( $(date +%s) - t0 ))s"; tail -c 600 $O/$name.out | tail -4
    if [ $rc -ne 0 ]; then tail -3 $O/$name.err; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its time limit: stopping the session"; exit $rc; fi
done
echo "session $TAG done"
