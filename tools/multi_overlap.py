#!/usr/bin/env python3
"""The in-library multi-device context rehearsed on one device (D2D copies stand in for ncclSend / ncclRecv): N asynchronous frames,
run under `rocprofv3 --kernel-trace`; --summarise DIR reads the trace and says how often the de-interleave of frame k (the end of its
gather, on the root's second stream) ran while a render kernel of frame k + 1 was running.   VERDICT r3 item 5."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--summarise" in sys.argv:
    d = sys.argv[sys.argv.index("--summarise") + 1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ren = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("rt_trace"))
    dei = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("rt_deinterleave"))
    inside = sum(1 for a, b in dei if any(s < a and b < e for s, e in ren))
    touching = sum(1 for a, b in dei if any(s < b and a < e for s, e in ren))
    print(json.dumps({"render_kernels": len(ren), "deinterleave_kernels": len(dei), "deinterleaves_wholly_inside_a_render_kernel": inside,
                      "deinterleaves_overlapping_a_render_kernel": touching,
                      "mean_render_us": round(sum(e - s for s, e in ren) / max(len(ren), 1) / 1e3, 1),
                      "mean_deinterleave_us": round(sum(e - s for s, e in dei) / max(len(dei), 1) / 1e3, 1)}))
    sys.exit(0)
from raytracing_simple_amd import api, host
w, h, spp = 1920, 1080, 16
sph = host.demo_scene()
cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
for n in (2, 4, 8):
    with api.RtContext(w, h, devices=[0] * n) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        for _ in range(24):
            ctx.reset_async()
            ctx.render_async(spp)
        ctx.read_pixels()
    print("shards", n, "done", flush=True)
