#!/bin/bash
# The round's final library against the oracle, beyond the suite (one gpurun call; output gpurun_out/<tag>/validation.txt):
#   tools/validation.sh TAG [SEEDS_PER_FAMILY] [FIRST_SEED]
# full-size parity of every BASELINE configuration; fuzzed scenes of four families with the library's own choice, with the hierarchy
# forced (tree shaped on the host) and with the hierarchy forced and shaped ON THE DEVICE (what updates and large uploads get);
# random API sequences (passes, resets, streams, shards, device-resident updates); adversarial rays through the walk and the sweep.
set -u
TAG=${1:-validation}; N=${2:-600}; S=${3:-70000}
O=gpurun_out/$TAG; mkdir -p $O; F=$O/validation.txt
echo "# validation, build $(python3 -c 'import sys; sys.path.insert(0, "."); from raytracing_simple_amd import api; print(api.build_id())')" > $F
echo "## tools/full_size_parity.py" >> $F
timeout -k 10 600 python tools/full_size_parity.py c1,c2,c16,c3,c5,c4 >> $F 2>&1; echo "full size done"
for fam in 1 2 3 4 5 6; do
    echo "## fuzz family $fam, $N seeds from $S: the library's choice / hierarchy forced (host shape) / hierarchy forced, shaped on the device" >> $F
    timeout -k 10 900 python tools/fuzz_parity.py $S $N $fam 2>&1 | tail -3 >> $F
    RT_FUZZ_BVH=1 timeout -k 10 900 python tools/fuzz_parity.py $((S + 10000)) $N $fam 2>&1 | tail -3 >> $F
    RT_FUZZ_BVH=1 RT_FUZZ_TREE_SHAPE=2 timeout -k 10 900 python tools/fuzz_parity.py $((S + 20000)) $N $fam 2>&1 | tail -3 >> $F
    echo "family $fam done"
done
echo "## tools/fuzz_api.py: random operation sequences" >> $F
timeout -k 10 900 python tools/fuzz_api.py $((S / 10)) $((N / 2)) 2>&1 | tail -3 >> $F; echo "api done"
echo "## tools/ray_campaign.py: adversarial rays through the walk and the plain sweep" >> $F
timeout -k 10 600 python tools/ray_campaign.py 300000 2>&1 | tail -14 >> $F
echo "validation done"; grep -c "mismatches: \[\]" $F; grep -i "MISMATCH seed\|differ': [1-9]\|false" $F | head
