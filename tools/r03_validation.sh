#!/bin/bash
# Round-3 validation campaign on the GPU box (through gpurun): the final library against the oracle beyond what the suite holds.
set -u
O=gpurun_out/${1:-r03v}; mkdir -p $O
python tools/full_size_parity.py c1,c2,c16,c3,c5,c4 > $O/full_size_parity.jsonl 2> $O/full_size.err; cut -c1-260 $O/full_size_parity.jsonl
for fam in 1 2 3 4; do python tools/fuzz_parity.py 30000 1200 $fam >> $O/fuzz.log 2>&1; done
for fam in 1 2 3 4; do RT_FUZZ_BVH=1 python tools/fuzz_parity.py 40000 1200 $fam >> $O/fuzz_bvh.log 2>&1; done
grep -h "fuzz seeds\|MISMATCH" $O/fuzz.log $O/fuzz_bvh.log | tail -12
python tools/fuzz_api.py 7000 400 > $O/fuzz_api.log 2>&1; tail -2 $O/fuzz_api.log
python tools/ray_campaign.py 300000 > $O/rays.log 2>&1; tail -2 $O/rays.log
echo "validation done"
