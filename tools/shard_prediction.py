#!/usr/bin/env python3
"""What ONE GPU can measure about N > 1: every shard of an N-way split rendered ALONE.

    python tools/shard_prediction.py [--workloads c2,c4] [--nranks 2,4,8] [--frames 3] [--out profiles/r06_shard_prediction.jsonl]

In an N-GPU run of the row-tile split (tile t -> GPU t mod N, 8-row tiles: include/rt_api.h rt_create_sharded) GPU r renders
exactly the launch that `rt_create_sharded(rank r, nranks N)` renders here -- same pixels, same seeds, same kernel instance, on
a GPU that holds nothing else.  Its steady-state kernel time on this GPU is therefore what that GPU would spend per frame, with
the two effects the perfect-balance bound of bench.py (`config.strong_scaling_bound`) leaves out: shards are not equally heavy,
and a shard of 1/N of the image leaves the GPU under-filled (at 8 shards a 1080p shard is 4 050 wavefronts on 6 144 resident
slots: one partial wave of occupancy, all tail).  A frame of the N-GPU run lasts

    predicted_ms_per_frame = max over shards (kernel ms) + de-interleave kernel on the root (measured) + gather of the packed rows

with the gather priced as the whole frame's packed pixels over ONE xGMI link (4 * w * h bytes / 153 GB/s: an upper estimate --
the root receives from N - 1 peers over distinct links at once, so (N-1)/N of that is spread over N - 1 links; both figures
are recorded).  Every shard's rows are compared with the unsharded frame's, bit for bit.

One JSON object per (workload, N) on stdout and, with --out, appended to that file; `predict()` is what bench.py calls for the
`config.predicted_from_shards` block of its line."""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TILE_ROWS = 8
WARM = 4                              # untimed frames per context: a scene of 4-11 spheres has its instance measured on frames 2 and 3 (rt_launch.hip launch_small)
XGMI_LINK_GBS = 153.0                 # MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point


def _deinterleave_ms(api, w, h, n, device=0, reps=20):
    """The root's frame-assembly kernel (rt_deinterleave_rows) on buffers of this frame's size, device time per call."""
    import torch
    dev = torch.device("cuda", device)
    pad = max(len(api.local_rows_of(h, r, n, TILE_ROWS)) for r in range(n))
    gathered = torch.zeros((n, pad, w), dtype=torch.int32, device=dev)
    full = torch.zeros((h, w), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        api.deinterleave_rows(full.data_ptr(), gathered.data_ptr(), w, h, n, TILE_ROWS, pad, device=device, stream=st.cuda_stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        api.deinterleave_rows(full.data_ptr(), gathered.data_ptr(), w, h, n, TILE_ROWS, pad, device=device, stream=st.cuda_stream)
    b.record(st)
    b.synchronize()
    return a.elapsed_time(b) / reps


def predict(api, mode, spheres, cam, w, h, spp, nranks=(2, 4, 8), frames=3, device=0, whole_ms=None, whole_pixels=None, bound=None):
    """[{n, shard_ms[], max, mean, imbalance, predicted_ms_per_frame, ...}] for the frame (spheres, cam, w, h, spp) in `mode`.
    `whole_ms` / `whole_pixels`: the unsharded frame's steady kernel time and pixels (rendered here when not given);
    `bound`: bench.py's strong_scaling_bound block, to put the perfect-balance figure beside the measured one."""
    if whole_ms is None or whole_pixels is None:
        with api.RtContext(w, h, device=device) as c:
            c.set_scene(spheres); c.set_camera(cam); c.set_mode(mode)
            ms = []
            for k in range(WARM + frames):
                c.reset()
                px = c.render_pass(spp)
                if k >= WARM:
                    ms.append(c.stats()["last_kernel_ms"])
            whole_ms, whole_pixels = statistics.median(ms), px
    whole = np.asarray(whole_pixels, np.uint32).reshape(h, w)
    out = []
    for n in nranks:
        shard_ms, rows_n, kernels, equal = [], [], set(), True
        for r in range(n):
            with api.RtContext(w, h, device=device, rank=r, nranks=n, tile_rows=TILE_ROWS) as c:
                c.set_scene(spheres); c.set_camera(cam); c.set_mode(mode)
                ms = []
                for k in range(WARM + frames):      # untimed frames first: tile costs, the heavy-first order, the library's own choice of instance (bench.py does the same)
                    c.reset()
                    px = c.render_pass(spp)
                    if k >= WARM:
                        ms.append(c.stats()["last_kernel_ms"])
                rows = c.local_row_map()
                equal = equal and bool(np.array_equal(px.reshape(-1, w)[: len(rows)], whole[rows]))
                shard_ms.append(statistics.median(ms))
                rows_n.append(int(len(rows)))
                kernels.add(c.last_kernel)
        mx, mean = max(shard_ms), sum(shard_ms) / n
        deint = _deinterleave_ms(api, w, h, n, device)
        gather_one_link = 4.0 * w * h / (XGMI_LINK_GBS * 1e9) * 1e3
        gather_parallel = 4.0 * w * h / n / (XGMI_LINK_GBS * 1e9) * 1e3       # each peer's 1/N of the frame over its own link, all at once
        pred = mx + deint + gather_one_link
        rec = {"n_gpus": n, "shard_kernel_ms": [round(v, 4) for v in shard_ms], "shard_rows": rows_n, "kernel": sorted(kernels),
               "max_ms": round(mx, 4), "mean_ms": round(mean, 4), "imbalance_max_over_mean": round(mx / mean, 4),
               "one_gpu_frame_ms": round(whole_ms, 4), "sum_of_shards_over_one_gpu_frame": round(sum(shard_ms) / whole_ms, 4),
               "deinterleave_ms": round(deint, 4), "gather_ms_all_bytes_over_one_link": round(gather_one_link, 4),
               "gather_ms_each_peer_on_its_own_link": round(gather_parallel, 4),
               "predicted_ms_per_frame": round(pred, 4), "predicted_speedup": round(whole_ms / pred, 3),
               "predicted_speedup_kernel_only": round(whole_ms / mx, 3),
               "every_shard_equals_unsharded_rows": equal}
        if bound and "slowest_wavefront_under_full_occupancy_ms" in bound:
            perfect = max(bound["slowest_wavefront_under_full_occupancy_ms"], whole_ms / n)
            rec["perfect_balance_bound_ms"] = round(perfect, 4)
            rec["predicted_over_perfect_balance_bound"] = round(pred / perfect, 3)
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,c4")
    ap.add_argument("--nranks", default="2,4,8")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--mode", choices=["parity", "fast"], default="parity")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch                        # (first: torch's HIP runtime must be the one the process initialises -- with the library's own context
    torch.cuda.init()                   #  created before it, torch.cuda reported "No HIP GPUs are available" on the GPU box)
    from raytracing_simple_amd import api, host
    from tools.ab_bench import CONFIGS
    import bench
    mode = api.RT_MODE_FAST if args.mode == "fast" else api.RT_MODE_PARITY
    for wl in args.workloads.split(","):
        maker, w, h, spp = CONFIGS[wl]
        sph, orig, target = maker()
        cam = host.compute_camera(orig, target, w, h)
        bound = bench.scaling_bound(api, mode, sph, cam, w, h, spp)
        recs = predict(api, mode, sph, cam, w, h, spp, nranks=[int(v) for v in args.nranks.split(",")], frames=args.frames, bound=bound)
        for rec in recs:
            line = json.dumps(dict({"workload": wl, "w": w, "h": h, "spp": spp, "mode": args.mode, "spheres": int(len(sph)), "tile_rows": TILE_ROWS,
                                    "build_id": api.build_id()}, **rec, slowest_wavefront_under_full_occupancy_ms=bound.get("slowest_wavefront_under_full_occupancy_ms")))
            print(line, flush=True)
            if args.out:
                with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "a") as f:
                    f.write(line + "\n")


if __name__ == "__main__":
    main()
