#!/usr/bin/env python3
"""Where the plain sweep should leave LDS: scenes without a usable hierarchy (forced off here) of 500 ... 9 500 records, rendered by the cooperative sweep
over tables staged in LDS (forced by name: at ever fewer workgroups per CU as the tables grow; the library's pick up to 40 KB) and by rt_trace_parity_g
forced by name (table through the scalar cache, six wavefronts per SIMD whatever the size; the library's pick beyond 40 KB).  1080p, 1 pass (2 for the small ones), median kernel ms of 3 frames, frames
compared bit for bit.  `python tools/g_threshold_probe.py` (diagnostics library) -> profiles/r06_g_threshold.jsonl"""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raytracing_simple_amd import api, host, scenes  # noqa: E402


def run(sph, cam, w, h, spp, inst):
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        if inst:
            ctx.set_mode(api.instance_mode(inst))
        ts, px = [], None
        for _ in range(4):
            ctx.reset()
            px = ctx.render_pass(spp)
            ts.append(ctx.stats()["last_kernel_ms"])
        return statistics.median(ts[1:]), ctx.last_kernel, px


def main():
    w, h = 1920, 1080
    for family, maker in (("Demo + NaN records", lambda n: bench.nan_scene(n)), ("closed box of mirror / glass spheres", lambda n: scenes.mirror_box(n))):
        for n in (500, 1000, 1500, 2000, 3000, 4000, 6000, 9500):
            if family.startswith("closed") and n > 4000:
                continue
            sph, orig, target = maker(n)
            cam = host.compute_camera(orig, target, w, h)
            spp = 1
            a_ms, a_k, a_px = run(sph, cam, w, h, spp, "rt_trace_parity_coop")
            b_ms, b_k, b_px = run(sph, cam, w, h, spp, "rt_trace_parity_g")
            lds = 16 * len(sph) + 64
            print(json.dumps({"family": family, "records": int(len(sph)), "w": w, "h": h, "spp": spp, "staged": a_k, "staged_ms": round(a_ms, 3),
                              "tables_in_lds_bytes": lds, "workgroups_per_cu": max(1, min(6, (160 * 1024) // (lds + 2048))),
                              "scalar_cache": b_k, "scalar_cache_ms": round(b_ms, 3), "scalar_cache_over_staged": round(b_ms / a_ms, 3),
                              "frames_equal": bool(np.array_equal(a_px, b_px))}), flush=True)


if __name__ == "__main__":
    main()
