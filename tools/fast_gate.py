#!/usr/bin/env python3
"""Fast mode against north_star's tolerance, per BASELINE configuration.

    python tools/fast_gate.py [--configs c2,c16,c3,c256,c4,c5] [--rounds 3] > profiles/rNN_fast_gate.jsonl

north_star: "PSNR >= 50 dB against [the reference CPU path] for multi-spp float accumulation".  Parity mode is bit-equal to
that path, so the figure is the PSNR of the fast frame against the parity frame of the same workload at its own sample
count.  Three frames per configuration, interleaved in ONE process (the diagnostics library):
    parity      what the library renders in RT_MODE_PARITY
    fast        what it renders in RT_MODE_FAST (fused multiply-adds, hardware rcp / rsq / sqrt / sin / cos / exp / log)
    fast_dx     the experiment: fast mode with every DECISION in parity arithmetic (RT_OPT_EXACT_DECISIONS, rt_trace.inc.h) --
                the instance of the same shape as the one fast mode chose (rt_trace_fastdx_*)
One JSON line per configuration."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402
from tools.ab_bench import CONFIGS  # noqa: E402

GATE_DB = 50.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c16,c3,c256,c4,c5")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    names = api.instance_names(fast=True)
    for cname in args.configs.split(","):
        maker, w, h, spp = CONFIGS[cname]
        sph, orig, target = maker()
        cam = host.compute_camera(orig, target, w, h)
        with api.RtContext(w, h, diag=True) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            arms = {"parity": api.RT_MODE_PARITY, "fast": api.RT_MODE_FAST}
            pix, ms, kern = {}, {}, {}
            for r in range(args.rounds + 1):
                for arm, mode in list(arms.items()):
                    ctx.set_mode(mode)
                    ctx.reset()
                    px = ctx.render_pass(spp)
                    st = ctx.stats()
                    kern[arm] = ctx.last_kernel
                    if r == 0:
                        pix[arm] = px
                        if arm == "fast":       # the experiment's instance has the shape fast mode chose for this scene
                            dx = kern["fast"].replace("rt_trace_fast", "rt_trace_fastdx")
                            if dx in names:
                                arms["fast_dx"] = api.instance_mode(dx)
                    else:
                        ms.setdefault(arm, []).append(st["last_kernel_ms"])
                if r == 0 and "fast_dx" in arms:            # (its warm-up frame: the arm joined during round 0)
                    ctx.set_mode(arms["fast_dx"])
                    ctx.reset()
                    pix["fast_dx"] = ctx.render_pass(spp)
                    kern["fast_dx"] = ctx.last_kernel
            rec = {"config": cname, "spheres": int(len(sph)), "w": w, "h": h, "spp": spp, "gate_db": GATE_DB, "build_id": api.build_id(diag=True)}
            for arm in arms:
                t = statistics.median(ms[arm])
                rec[arm] = {"kernel": kern[arm], "kernel_ms": round(t, 4)}
                if arm != "parity":
                    db = host.psnr(pix[arm], pix["parity"])
                    rec[arm].update({"psnr_db_vs_parity": round(db, 2), "meets_north_star_gate": bool(db >= GATE_DB),
                                     "ms_vs_parity": round(t / statistics.median(ms["parity"]), 3)})
            if "fast_dx" in arms:
                rec["fast_dx"]["ms_vs_fast"] = round(rec["fast_dx"]["kernel_ms"] / rec["fast"]["kernel_ms"], 3)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
