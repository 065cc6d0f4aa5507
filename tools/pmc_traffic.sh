#!/bin/bash
# HBM traffic of the kernel a configuration renders with: tools/pmc_traffic.sh CONFIG   (FETCH_SIZE and WRITE_SIZE in separate passes;
# RT_DEAL / RT_NO_DEAL select the deal of pixels by cost, tools/pmc_modes.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}
OUT=$R/gpurun_out/pmc_traffic_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/f.log" 2>&1 || { tail -5 "$OUT/f.log"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 $R/tools/pmc_modes.py $CFG 0 > "$OUT/w.log" 2>&1 || { tail -5 "$OUT/w.log"; exit 1; }
grep MODE "$OUT/f.log" | cut -c1-220
python3 $R/tools/pmc_last.py "$OUT"
