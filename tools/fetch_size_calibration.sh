#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts in the render kernels' access shapes (tools/ubench/fetch_size.hip), two separate
# --pmc passes; the summary goes to gpurun_out/TAG/fetch_size_calibration.json (committed as profiles/r06_fetch_size_calibration.json).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-fetch_cal}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
BIN=$R/tools/ubench/fetch_size
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/ubench/fetch_size.hip -o $BIN || exit 1
$BIN > "$OUT/bytes.json" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- $BIN > "$OUT/pmc_$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$OUT/pmc_$c.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
nbytes = json.load(open(os.path.join(out, "bytes.json")))["bytes"]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        per.setdefault((k, int(r["Dispatch_Id"])), 0.0)
        per[(k, int(r["Dispatch_Id"]))] += float(r["Counter_Value"]) * 1024.0      # (the counters' unit is KiB)
    for (k, _), v in per.items():
        res.setdefault(k, {}).setdefault(c, []).append(v)
summary = {}
for k, d in res.items():
    b = nbytes.get(k)
    if not b:
        continue
    row = {"bytes_moved": b}
    for c, vals in d.items():
        row[c + "_bytes_per_launch"] = [round(v) for v in vals]
        row[c + "_over_bytes"] = round(sum(vals) / len(vals) / b, 4)
    summary[k] = row
json.dump(summary, open(os.path.join(out, "fetch_size_calibration.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
