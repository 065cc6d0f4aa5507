#!/usr/bin/env python3
"""Where a launch's time goes at wavefront granularity: renders one frame with the wall-clock-logging instance
(instance rt_trace_parity_tl of librt_hip_diag.so: the shipped kernel + one s_memrealtime pair per wavefront) and reports the
resident-wavefront curve over the launch (ramp, plateau, tail), per-wavefront durations by image region and by
XCD, and how much of the launch the last wavefronts account for.
    python tools/wave_timeline.py [c2|c16|c5] [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
w, h, spp = 1920, 1080, 64
if wl == "c2":
    sph, orig, target = host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET
elif wl == "c16":
    sph, orig, target = scenes.demo_plus(16)
else:
    sph, orig, target = scenes.mirror_box(64)
cam = host.compute_camera(orig, target, w, h)
lib = api.load_library(diag=True)
gx, gy = (w + 31) // 32, (h + 7) // 8
n_waves = gx * gy * 4
with api.RtContext(w, h, diag=True) as ctx:
    ctx.set_scene(sph); ctx.set_camera(cam)
    ctx.set_mode(api.instance_mode("rt_trace_parity" if len(sph) < 12 else "rt_trace_parity_coop"))
    for _ in range(3):
        ctx.reset(); ctx.render_pass(spp)
    plain_ms = ctx.stats()["last_kernel_ms"]
    ctx.set_mode(api.instance_mode("rt_trace_parity_tl"))
    api._check(lib.rt_debug_timelog_enable(ctx._h, 8, n_waves), lib)
    ctx.reset(); ctx.render_pass(spp)
    logged_ms = ctx.stats()["last_kernel_ms"]
    buf = np.zeros(n_waves * 3, np.uint64)
    api._check(lib.rt_debug_wavelog_read(ctx._h, buf.ctypes.data_as(C.c_void_p), n_waves), lib)
rec = buf.reshape(n_waves, 3)
start, end = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 15
hwid = (rec[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu, se, simd = (hwid >> 8) & 15, (hwid >> 13) & 7, (hwid >> 4) & 3
t0, t1 = start.min(), end.max()
span = float(t1 - t0)                                   # 10 ns ticks
dur = (end - start).astype(np.float64)
bins = 60
edges = np.linspace(t0, t1, bins + 1)
resident = [int(((start < edges[i + 1]) & (end > edges[i])).sum()) for i in range(bins)]
mid = [float(((np.minimum(end, edges[i + 1]) - np.maximum(start, edges[i])).clip(0)).sum() / (edges[i + 1] - edges[i])) for i in range(bins)]
wave_ticks = float(dur.sum())
by_row = {}
block_row = (np.arange(n_waves) // 4) // gx
for lo, hi, name in ((0, gy // 4, "bottom quarter"), (gy // 4, gy // 2, "second"), (gy // 2, 3 * gy // 4, "third"), (3 * gy // 4, gy, "top quarter")):
    m = (block_row >= lo) & (block_row < hi)
    by_row[name] = {"waves": int(m.sum()), "mean_us": float(dur[m].mean() / 100), "mean_start_us": float((start[m] - t0).mean() / 100),
                    "mean_end_us": float((end[m] - t0).mean() / 100)}
order = np.argsort(end)
last = order[-int(0.02 * n_waves):]
res = {
    "workload": wl, "kernel_ms_plain": plain_ms, "kernel_ms_logged": logged_ms, "span_ms": span / 1e5, "waves": n_waves,
    "wave_us": {"mean": float(dur.mean() / 100), "p10": float(np.percentile(dur, 10) / 100), "p50": float(np.percentile(dur, 50) / 100),
                "p90": float(np.percentile(dur, 90) / 100), "max": float(dur.max() / 100)},
    "mean_resident_waves": wave_ticks / span, "slots": 6144,
    "resident_curve_avg": [round(v) for v in mid],
    "time_with_less_than_half_the_slots_filled_ms": float(sum((edges[i + 1] - edges[i]) for i in range(bins) if mid[i] < 3072) / 1e5),
    "last_2pct_waves": {"mean_start_us": float((start[last] - t0).mean() / 100), "mean_us": float(dur[last].mean() / 100),
                        "block_rows": [int(v) for v in np.unique(block_row[last])[:12]]},
    "by_image_region": by_row,
    "per_xcc": {int(x): {"waves": int((xcc == x).sum()), "last_end_us": float((end[xcc == x].max() - t0) / 100),
                         "wave_ticks_share": float(dur[xcc == x].sum() / wave_ticks)} for x in np.unique(xcc)},
    "distinct_cu_ids": int(len(np.unique(xcc * 1000 + se * 100 + cu))),
}
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
