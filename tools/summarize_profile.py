#!/usr/bin/env python3
"""Summarise gpurun_out/TAG (written by tools/profile_gpu.sh) into profiles/: the kernel-trace
stats table and per-launch PMC averages for the render kernel, with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced reads; its unit and
WRITE_SIZE's are KiB)."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def main():
    tag, mode, out_name = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", tag)
    kern = sys.argv[4] if len(sys.argv) > 4 else "rt_trace_" + mode + "_w1"      # the C2 instance: single-wavefront workgroups
    summary = {"mode": mode, "kernel": kern, "source": f"gpurun_out/{tag} (tools/profile_gpu.sh)"}
    # the library the counters were measured on (written on the GPU box by tools/profile_gpu.sh: rt_build_id() = a hash of
    # csrc/, the public headers and the compiler flags); bench.py prints these figures only beside that very library
    try:
        summary["build_id"] = open(os.path.join(src, "build_id.txt")).read().strip()
    except OSError:
        summary["build_id"] = None
    stats = find(os.path.join(src, "trace"), "*kernel_stats.csv")
    lines = []
    if stats:
        with open(stats) as f:
            rows = list(csv.DictReader(f))
        lines.append("| kernel | calls | avg ns, ALL launches (pricing and probe launches of a few passes included: compare `avg_kernel_ns_timed_launches` below with ms_per_step) | min ns | max ns | % |")
        lines.append("|---|---|---|---|---|---|")
        for r in rows:
            lines.append(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
            if r["Name"] == kern:
                summary["avg_kernel_ns"] = float(r["AverageNs"])
                summary["calls"] = int(r["Calls"])
    # the timed launches = the last RT_PROF_LAST dispatches of the kernel (bench.py --steps; the ones before them are the
    # untimed frames that leave tile costs and the heavy-first order)
    last_n = int(os.environ.get("RT_PROF_LAST", "10"))
    trace = find(os.path.join(src, "trace"), "*kernel_trace.csv")
    if trace:
        with open(trace) as f:
            rows = [r for r in csv.DictReader(f) if r["Kernel_Name"] == kern]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        tail = rows[-last_n:]
        if tail:
            durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail]
            summary["timed_launches"] = len(tail)
            summary["avg_kernel_ns_timed_launches"] = sum(durs) / len(durs)
            summary["min_kernel_ns_timed_launches"] = min(durs)
        with open(trace) as f:
            for r in csv.DictReader(f):
                if r["Kernel_Name"] == kern:
                    summary["launch"] = {k: r[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                           "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y")}
                    break
    counters = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        f = find(d, "*counter_collection.csv")
        if not f:
            continue
        with open(f) as fh:
            rows = [r for r in csv.DictReader(fh) if r.get("Kernel_Name") == kern]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last_n:]
        per = {}
        for r in rows:                      # (a counter is reported once per XCD / SE instance of a dispatch: sum those, average the dispatches)
            if int(r["Dispatch_Id"]) in ids:
                per.setdefault((r["Counter_Name"], int(r["Dispatch_Id"])), 0.0)
                per[(r["Counter_Name"], int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
        for (name, _), v in per.items():
            counters.setdefault(name, []).append(v)
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    summary["pmc_per_launch_avg"] = avg
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        fetch_b = avg["FETCH_SIZE"] * 1024.0
        write_b = avg["WRITE_SIZE"] * 1024.0
        summary["hbm_fetch_bytes_raw"] = fetch_b
        summary["hbm_write_bytes"] = write_b
        # guide: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream on gfx950 -> upper bound 2x
        summary["hbm_bytes_per_launch"] = fetch_b + write_b
        summary["hbm_bytes_per_launch_fetch_x2"] = 2 * fetch_b + write_b
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        summary["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"], 1.0)
    if "SQ_INSTS_VALU" in avg and "SQ_THREAD_CYCLES_VALU" in avg:
        # SQ_INSTS_VALU is an exact instruction count; SQ_THREAD_CYCLES_VALU counts active lanes per instruction
        summary["active_lane_frac"] = avg["SQ_THREAD_CYCLES_VALU"] / (avg["SQ_INSTS_VALU"] * 64.0)
        summary["valu_insts_per_launch"] = avg["SQ_INSTS_VALU"]
    if "SQ_INSTS_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        # issue cycles per SIMD: a wave64 fp32 VALU instruction occupies its SIMD for 2 cycles, a
        # transcendental for 8, an f64 operation for 4 (tools/ubench/valu_rate.hip); 1024 SIMDs;
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        trans = avg.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        f64 = sum(avg.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
        busy = (2.0 * avg["SQ_INSTS_VALU"] + 6.0 * trans + 2.0 * f64) / 1024.0
        summary["valu_busy_frac_single_stream"] = busy / (avg["GRBM_GUI_ACTIVE"] / 8.0)
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    with open(os.path.join(root, "profiles", out_name + ".json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    with open(os.path.join(root, "profiles", out_name + ".md"), "w") as f:
        label = sys.argv[5] if len(sys.argv) > 5 else "C2: Demo, 1920x1080, 64 spp"
        f.write(f"# {out_name}: rocprofv3 summary, bench.py --mode {mode} ({label})\n\n")
        f.write("## --kernel-trace --stats\n\n" + "\n".join(lines) + "\n\n")
        f.write("## launch\n\n```\n" + json.dumps(summary.get("launch", {}), indent=1) + "\n```\n\n")
        f.write("## PMC, average per launch of the render kernel (separate passes)\n\n| counter | value |\n|---|---|\n")
        for k in sorted(avg):
            f.write(f"| {k} | {avg[k]:.6g} |\n")
        f.write("\n## derived\n\n```\n" + json.dumps({k: v for k, v in summary.items()
                                                       if k not in ("pmc_per_launch_avg", "launch")}, indent=1) + "\n```\n")
    # RT_PMC_KEY=c16 (the bench.py workload the profile is of): also the record bench.py reads for that workload's
    # `roofline.traffic` / `roofline.executed` (profiles/pmc_traffic.json)
    key = os.environ.get("RT_PMC_KEY")
    if key and "valu_insts_per_launch" in summary and "avg_kernel_ns" in summary:
        pmc_path = os.path.join(root, "profiles", "pmc_traffic.json")
        try:
            data = json.load(open(pmc_path))
        except (OSError, ValueError):
            data = {}
        rec = {"kernel": kern, "build_id": summary.get("build_id"), "valu_insts_per_launch": summary["valu_insts_per_launch"], "active_lane_frac": round(summary["active_lane_frac"], 4),
               "profiled_kernel_ms": round(summary.get("avg_kernel_ns_timed_launches", summary["avg_kernel_ns"]) / 1e6, 4),
               "source": f"profiles/{out_name}.{{md,json}} (rocprofv3 --kernel-trace --stats and separate --pmc passes of `bench.py --workload {key} --no-extras`, tools/profile_gpu.sh)"}
        if "valu_busy_frac_single_stream" in summary:
            rec["valu_busy_frac_single_stream"] = round(summary["valu_busy_frac_single_stream"], 4)
        if "hbm_bytes_per_launch" in summary:
            rec.update({"hbm_bytes_per_launch": int(summary["hbm_bytes_per_launch"]), "hbm_bytes_per_launch_fetch_x2": int(summary["hbm_bytes_per_launch_fetch_x2"]),
                        "fetch_raw_bytes": round(summary["hbm_fetch_bytes_raw"], 1), "write_bytes": round(summary["hbm_write_bytes"], 1)})
        old = (data.get(key) or {}).get(mode) or {}
        if old.get("build_id") == rec["build_id"] and "l2_hit_rate_staged_tables" in old:     # (tools/pmc_staging.sh of the same library)
            rec["l2_hit_rate_staged_tables"] = old["l2_hit_rate_staged_tables"]
        if "l2_hit_rate" in summary:
            rec["l2_hit_rate"] = round(summary["l2_hit_rate"], 4)
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_frac"] = round(avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"], 4)
        data.setdefault(key, {})[mode] = rec
        with open(pmc_path, "w") as f:
            json.dump(data, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
