#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

A step = ONE FRAME of BASELINE.json configs[1]: Demo scene (6 spheres), 1920x1080, 64 samples
per pixel (= 64 passes of the reference's Config::updateRendering()), from the default seed
stream, rendered by the HIP path through the C ABI.  Inputs (seeds, scene tables, camera) are
resident in HBM before the timed region.  With N > 1 (one process per GPU, launched by
torch.distributed.run) the image is sharded by interleaved 8-row tiles and each frame ends with
one RCCL gather of the packed pixels to rank 0 (issued asynchronously: frame k's gather overlaps
frame k+1's render, two send buffers); total work is fixed, so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode parity|fast] [--no-cpu]

Prints ONE JSON line (rank 0).  `value` = rays (primary + shadow) of all ranks / wall time of the
K timed steps (max over ranks), in Mray/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP = 1920, 1080, 64
TILE_ROWS = 8
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0                # same table
FLOP_PER_SPHERE_TEST = 20            # ray-sphere test
BYTES_PER_PIXEL_PER_LAUNCH = 32      # seeds 8 R + 8 W, colour 12 W, pixel 4 W


def cpu_baseline(spheres, cam):
    """The oracle (CPU port of the reference kernel) on the host cores, same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = os.cpu_count() or 1
    t0 = time.time()
    out = O.render(spheres, cam, W, H, SPP, threads=cores)
    dt = time.time() - t0
    st = out["stats"]
    rays = st["samples"] + st["shadow_calls"]
    return {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"full workload: {W}x{H} x {SPP} spp, {rays} rays in {dt:.2f} s "
                      f"({st['samples'] / dt / 1e6:.1f} Msample/s)",
            "ms_per_frame": round(dt * 1e3, 1)}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["parity", "fast"], default="parity")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch                       # plumbing: streams, events, torch.distributed (RCCL)
    import torch.distributed as dist

    from raytracing_simple_amd import api, host
    from raytracing_simple_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # rehearsal knobs for a one-GPU box (never set by the driver): every rank on device 0,
    # gloo instead of RCCL
    if os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    spheres = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W, H)
    mode = api.RT_MODE_FAST if args.mode == "fast" else api.RT_MODE_PARITY

    ctx = api.RtContext(W, H, device=local_rank, rank=rank, nranks=world, tile_rows=TILE_ROWS)
    ctx.set_scene(spheres)
    ctx.set_camera(cam)
    ctx.set_mode(mode)

    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    gather = None
    SLOTS = 2          # frame k's gather overlaps frame k+1's render
    if world > 1:
        gather = rdist.FrameGatherer(H, W, rank, world, TILE_ROWS, dev, slots=SLOTS)
    frame_no = [0]

    def step(ev=None):
        k = frame_no[0]
        frame_no[0] += 1
        if gather is not None:
            gather.wait(k)                                  # slot free again (its gather of frame k-2)
            buf = gather.local_slot(k)
            ctx.set_pixel_buffer(buf.data_ptr(), buf.numel())   # render straight into the send buffer
        ctx.reset_async(sh)
        if ev:
            ev[0].record(stream)
        ctx.render_async(SPP, sh)
        if ev:
            ev[1].record(stream)
        if gather is not None:
            gather.gather(k, async_op=True)                 # queued behind the launch, not waited for

    def drain():
        if gather is not None:
            for k in range(SLOTS):
                gather.wait(k)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    drain()                                                 # every frame gathered and assembled
    sync()
    elapsed = time.perf_counter() - t0

    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(args.steps, 1)

    # the other arithmetic mode on the same workload, outside the headline's timed region
    # (N = 1 only; reported beside `value`, never instead of it)
    other = None
    if world == 1:
        other_mode = api.RT_MODE_FAST if mode == api.RT_MODE_PARITY else api.RT_MODE_PARITY
        last_pixels = ctx.render_pass(0, copy=True)           # frame of the last timed step
        ctx.set_mode(other_mode)
        n_other = 8
        for _ in range(2):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(n_other):
            step()
        sync()
        dt = (time.perf_counter() - t1) / n_other
        st_o = ctx.stats()
        px_o = ctx.render_pass(0, copy=True)
        other = {"mode": "fast" if other_mode == api.RT_MODE_FAST else "parity",
                 "ms_per_step": round(dt * 1e3, 4),
                 "value": round((st_o["samples"] + st_o["shadow_rays"]) / dt / 1e6, 1), "unit": "Mray/s",
                 "psnr_db_vs_headline_mode": round(host.psnr(px_o, last_pixels), 2)}
        ctx.set_mode(mode)
        step()                                                # counters and frame of the headline mode again
        sync()
    frame_ok = None
    if world > 1 and rank == 0:
        # the gathered frame of the last step against an unsharded render on this GPU
        with api.RtContext(W, H, device=local_rank) as whole:
            whole.set_scene(spheres)
            whole.set_camera(cam)
            whole.set_mode(mode)
            want = whole.render_pass(SPP)
        got = gather.wait(frame_no[0] - 1).cpu().numpy().astype("uint32").reshape(-1)
        frame_ok = bool((got == want).all())
    st = ctx.stats()                      # counters of the last frame (reset clears them)
    counts = torch.tensor([st["samples"], st["closest_rays"], st["shadow_rays"], st["sphere_tests"]],
                          dtype=torch.int64, device=dev)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    k_max = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(k_max, op=dist.ReduceOp.MAX)
    samples, closest, shadow, tests = (int(v) for v in counts.tolist())
    elapsed = float(t_max.item())
    kernel_ms_max = float(k_max.item())

    if rank == 0:
        rays = samples + shadow                      # primary + shadow, the metric's ray count
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        # roofline of the dominant (only) kernel, per launch, from this rank's launches
        my_tests = st["sphere_tests"]
        flops = FLOP_PER_SPHERE_TEST * my_tests
        achieved_tflops = flops / (kernel_ms * 1e-3) / 1e12
        my_pixels = ctx.local_rows * W
        alg_bytes = BYTES_PER_PIXEL_PER_LAUNCH * my_pixels + 16 * len(spheres) * 3 + 60
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get(args.mode, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        line = {
            "metric": "Mray/s (primary+shadow) at 1080p 64spp",
            "value": round(value, 1),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C2: Demo scene (6 spheres), {W}x{H}, {SPP} spp, default seed stream",
                       "mode": args.mode, "collective": None if world == 1 else f"gather to rank 0 ({backend})",
                       "gathered_frame_equals_unsharded": frame_ok, "sharding": f"interleaved {TILE_ROWS}-row tiles x {world}",
                       "rays_per_frame": rays, "all_rays_per_frame": closest + shadow,
                       "Mray_s_all_rays": round((closest + shadow) * args.steps / elapsed / 1e6, 1),
                       "Msample_s": round(samples * args.steps / elapsed / 1e6, 1)},
            "roofline": {
                "bound": "valu-fp32",
                "kernel": "rt_trace_" + args.mode,
                "achieved": round(achieved_tflops, 3),
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, 5),
                "traffic": traffic,
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_max_rank": round(kernel_ms_max, 4),
                "algorithmic_flops_per_launch": flops,
                "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                        "achieved": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
            },
        }
        if other is not None:
            line["other_mode"] = other
        if world == 1 and not args.no_cpu:
            base, cpu_out = cpu_baseline(spheres, cam)
            line["cpu_baseline"] = base
            if args.mode == "parity":
                import numpy as np
                px = ctx.render_pass(0, copy=True)           # frame of the last timed step
                line["config"]["matches_cpu_oracle_bit_exact"] = bool(np.array_equal(px, cpu_out["pixels"]))
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
