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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# Before HIP starts: a hardware queue of its own for every stream in the process (frames in flight, the
# collective's streams, torch's).  With the default of 4 two frames shared a queue and did not overlap
# at all; and whenever streams SHARED queues (4, 8 or 16 queues for the ten-odd streams of a rank) the
# multi-rank frame loop with a torch.distributed collective in the process showed its ordering failure
# (tools/gather_stress.py); with 24 it never did, and nothing runs slower.

W, H, SPP = 1920, 1080, 64            # the headline workload (BASELINE.json configs[1]); --workload changes them
TILE_ROWS = 8
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0                # same table
FLOP_PER_SPHERE_TEST = 20            # ray-sphere test
BYTES_PER_PIXEL_PER_LAUNCH = 32      # seeds 8 R + 8 W, colour 12 W, pixel 4 W


def host_cores():
    """Cores this process may actually use: the smaller of the CPU count, the affinity mask and the
    cgroup CPU quota (a one-GPU box of the pool shows 256 CPUs and grants 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(spheres, cam):
    """The reference's CPU path on the host cores, same workload, same run: the reference's own
    kernel compiled as host C++ (oracle/_ref, kind "reference") when that build travelled with the
    snapshot, and the oracle (the CPU restatement, kind "port") -- the headline is the reference
    build when it is there.  Both are checkers: neither is part of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = host_cores()
    t0 = time.time()
    out = O.render(spheres, cam, W, H, SPP, threads=cores)
    dt = time.time() - t0
    st = out["stats"]
    rays = st["samples"] + st["shadow_calls"]
    where = f"on {cores} threads (host shows {os.cpu_count()} CPUs; cgroup quota / affinity grant {cores})"
    port = {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"full workload: {W}x{H} x {SPP} spp, {rays} rays in {dt:.2f} s "
                      f"({st['samples'] / dt / 1e6:.1f} Msample/s) {where}",
            "ms_per_frame": round(dt * 1e3, 1)}
    if not O.ref_available():
        return port, out
    import numpy as np
    t0 = time.time()
    ref = O.ref_render_mt(spheres, cam, W, H, SPP, cores)
    dt = time.time() - t0
    base = {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "reference",
            "sample": f"the reference's kernel source compiled as host C++ (oracle/_ref), full workload: {W}x{H} x {SPP} "
                      f"passes, {rays} rays in {dt:.2f} s {where}",
            "ms_per_frame": round(dt * 1e3, 1),
            "equals_port_bit_exact": bool(np.array_equal(ref["pixels"], out["pixels"])),
            "port": {k: port[k] for k in ("value", "ms_per_frame", "kind")}}
    return base, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["parity", "fast"], default="parity")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--workload", choices=["c2", "c16", "c3", "c4", "c5"], default="c2",
                    help="c2 = the headline (default; what the driver measures); the other BASELINE configurations on request")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frames kept in flight per rank (0 = 2 for N<=2, 3 for N<=4, 6 beyond)")
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

    global W, H, SPP
    from raytracing_simple_amd import scenes
    workloads = {
        "c2": ("C2: Demo scene (6 spheres)", lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 1920, 1080, 64),
        "c16": ("north-star target scene: Demo + 10 spheres (16)", lambda: scenes.demo_plus(16), 1920, 1080, 64),
        "c3": ("C3: 1024 random spheres", lambda: scenes.random_spheres(1024), 1920, 1080, 16),
        "c4": ("C4: Demo scene (6 spheres)", lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 3840, 2160, 256),
        "c5": ("C5: 64-sphere mirror box, depth 8", lambda: scenes.mirror_box(64), 1920, 1080, 64),
    }
    wl_name, wl_maker, W, H, SPP = workloads[args.workload]
    spheres, cam_orig, cam_target = wl_maker()
    cam = host.compute_camera(cam_orig, cam_target, W, H)
    mode = api.RT_MODE_FAST if args.mode == "fast" else api.RT_MODE_PARITY

    # Frames in flight.  A step is one frame; the K timed frames are independent (each restarts
    # from the default seed stream), so several can be in flight on separate streams, each with
    # its own seed/colour state -- what a renderer serving several views does.  One wavefront
    # needs about a millisecond for its 64 pixels x 64 spp: a single frame leaves the machine
    # partly idle while its last wavefronts finish (about 15 % at N = 1), and a 1/8 shard (4050
    # wavefronts for 6144 wavefront slots) cannot fill a GPU at all.  F frames in flight fill
    # those holes.  The single-stream figure is measured too (N = 1) and is the one the roofline
    # and the rocprof summaries refer to.
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if world <= 2 else (3 if world <= 4 else 6))
    ctxs = []
    for _ in range(F):
        c = api.RtContext(W, H, device=local_rank, rank=rank, nranks=world, tile_rows=TILE_ROWS)
        c.set_scene(spheres)
        c.set_camera(cam)
        c.set_mode(mode)
        ctxs.append(c)
    ctx = ctxs[0]

    main_stream = torch.cuda.current_stream()
    # each context's own HIP stream, wrapped for torch: they are created back to back and land on
    # distinct hardware queues, which streams from torch's pool did not always do (two pool streams
    # on one queue = no overlap at all; measured)
    side_streams = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    gather = None
    if world > 1:
        gather = rdist.FrameGatherer(H, W, rank, world, TILE_ROWS, dev, slots=2 * F)
        torch.cuda.synchronize()        # its zero-fills ran on torch's stream; the contexts' streams are non-blocking
        gather.gather(0)                # plumbing, not a step: RCCL builds its communicator and point-to-point
        torch.cuda.synchronize()        # channels on first use (seconds); keep that out of the timed region even with --warmup 0
        dist.barrier()
    frame_no = [0]

    def step(in_flight, ev=None, do_gather=True):
        k = frame_no[0]
        frame_no[0] += 1
        c = ctxs[k % in_flight]
        st = main_stream if in_flight == 1 else side_streams[k % in_flight]
        with torch.cuda.stream(st):
            if gather is not None and do_gather:
                gather.wait(k)                                  # slot free again (its last gather)
                buf = gather.local_slot(k)
                c.set_pixel_buffer(buf.data_ptr(), buf.numel())   # render straight into the send buffer
            c.reset_async(st.cuda_stream)
            if ev:
                ev[0].record(st)
            c.render_async(SPP, st.cuda_stream)
            if ev:
                ev[1].record(st)
            if gather is not None and do_gather:
                gather.gather(k, async_op=True)                 # queued behind the launch, not waited for
        return c

    def drain():
        if gather is not None:
            for k in range(2 * F):
                gather.wait(k)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(in_flight, steps, warmup, with_events=True, do_gather=True):
        """W untimed + exactly `steps` timed frames; barrier + synchronize on both sides.
        Per-launch HIP events only where asked: an event pair around every launch costs the
        overlapped region its overlap (measured), and a per-launch duration means little there."""
        for _ in range(warmup):
            step(in_flight, None, do_gather)
        drain()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(steps)] if with_events else None
        sync()
        t0 = time.perf_counter()
        last = None
        for k in range(steps):
            last = step(in_flight, events[k] if events else None, do_gather)
        drain()                                             # every frame gathered and assembled
        sync()
        elapsed = time.perf_counter() - t0
        kernel_ms = (sum(a.elapsed_time(b) for a, b in events) / max(steps, 1)) if events else elapsed / max(steps, 1) * 1e3
        return elapsed, kernel_ms, last

    single = None
    if world == 1:
        el1, k1, last1 = timed_region(1, args.steps, args.warmup)
        single = {"elapsed": el1, "kernel_ms": k1, "ctx": last1}
    else:
        # per-launch duration of this rank's shard for the roofline object: a few launches one at a
        # time, no gather, outside the headline's timed region
        _, k1, _ = timed_region(1, min(args.steps, 8), 1, with_events=True, do_gather=False)
        shard_kernel_ms = k1
    if F == 1 and single is not None:
        elapsed, kernel_ms, last_ctx = single["elapsed"], single["kernel_ms"], single["ctx"]
    else:
        elapsed, kernel_ms, last_ctx = timed_region(F, args.steps, args.warmup, with_events=False)
        if world > 1:
            kernel_ms = shard_kernel_ms

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

    last_pixels = last_ctx.render_pass(0, copy=True) if world == 1 else None   # the last TIMED frame (checked against the oracle below)

    def frame_counters(c):
        """Exact ray / test counts of one frame of this rank (the same for every frame): from one more,
        synchronous render after a synchronous reset, so that they do not depend on how the timed
        region's asynchronous resets were ordered."""
        c.set_pixel_buffer(0, 0)
        c.reset()
        c.render_pass(SPP, copy=False)
        return c.stats()

    st = frame_counters(last_ctx)

    # the other arithmetic mode on the same workload, single stream, outside the headline's
    # timed region (N = 1 only; reported beside `value`, never instead of it)
    other = None
    if world == 1:
        other_mode = api.RT_MODE_FAST if mode == api.RT_MODE_PARITY else api.RT_MODE_PARITY
        for c in ctxs:
            c.set_mode(other_mode)
        el_o, _, last_o = timed_region(F, 8, 2, with_events=False)
        px_o = last_o.render_pass(0, copy=True)
        st_o = frame_counters(last_o)
        other = {"mode": "fast" if other_mode == api.RT_MODE_FAST else "parity", "frames_in_flight": F,
                 "ms_per_step": round(el_o / 8 * 1e3, 4),
                 "value": round((st_o["samples"] + st_o["shadow_rays"]) * 8 / el_o / 1e6, 1), "unit": "Mray/s",
                 "psnr_db_vs_headline_mode": round(host.psnr(px_o, last_pixels), 2)}
        for c in ctxs:
            c.set_mode(mode)

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
        # roofline of the dominant (only) kernel, per launch, from this rank's launches; at N = 1
        # from the single-stream region, where launches do not overlap (what rocprof sees too)
        roof_kernel_ms = single["kernel_ms"] if single is not None else kernel_ms
        my_tests = st["sphere_tests"]
        flops = FLOP_PER_SPHERE_TEST * my_tests
        achieved_tflops = flops / (roof_kernel_ms * 1e-3) / 1e12
        my_pixels = ctx.local_rows * W
        alg_bytes = BYTES_PER_PIXEL_PER_LAUNCH * my_pixels + 16 * len(spheres) * 3 + 60
        traffic, executed = None, None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and args.workload == "c2" and os.path.exists(prof):
            try:
                pm = json.load(open(prof)).get(args.mode, {})
                traffic = pm.get("hbm_bytes_per_launch")
                if pm.get("valu_insts_per_launch"):
                    # what the VALU actually issues (PMC of the committed profile, same command): the
                    # time its instructions need at full issue rate, and how much of a step that is
                    floor_ms = pm["valu_busy_frac_single_stream"] * pm["profiled_kernel_ms"]
                    executed = {"valu_insts_per_launch": pm["valu_insts_per_launch"],
                                "active_lane_frac": pm["active_lane_frac"],
                                "valu_issue_floor_ms": round(floor_ms, 4),
                                "valu_busy_frac_one_frame_at_a_time": round(floor_ms / roof_kernel_ms, 4),
                                "valu_busy_frac_headline": round(floor_ms / ms_per_step, 4),
                                "l2_hit_rate": pm.get("l2_hit_rate"),
                                "source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes, tools/profile_gpu.sh)"}
            except (OSError, ValueError, KeyError):
                traffic, executed = None, None
        line = {
            "metric": "Mray/s (primary+shadow) at 1080p 64spp" if args.workload == "c2"
                      else f"Mray/s (primary+shadow) at {W}x{H} {SPP}spp",
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
            "config": {"workload": f"{wl_name}, {W}x{H}, {SPP} spp, default seed stream",
                       "mode": args.mode, "collective": None if world == 1 else f"gather to rank 0 ({backend})",
                       "frames_in_flight": F,
                       "gathered_frame_equals_unsharded": frame_ok, "sharding": f"interleaved {TILE_ROWS}-row tiles x {world}",
                       "rays_per_frame": rays, "all_rays_per_frame": closest + shadow,
                       "Mray_s_all_rays": round((closest + shadow) * args.steps / elapsed / 1e6, 1),
                       "Msample_s": round(samples * args.steps / elapsed / 1e6, 1)},
            "roofline": {
                "bound": "valu-fp32",
                "kernel": "rt_trace_" + args.mode + ("_coop" if len(spheres) >= 12 else ""),
                "achieved": round(achieved_tflops, 3),
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, 5),
                "traffic": traffic,
                "launches_overlap": False,
                "kernel_ms": round(roof_kernel_ms, 4),
                "step_ms_max_rank": round(kernel_ms_max, 4),
                "algorithmic_flops_per_launch": flops,
                "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                        "achieved": round(alg_bytes / (roof_kernel_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg_bytes / (roof_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
            },
        }
        if executed is not None:
            line["roofline"]["executed"] = executed
        if single is not None and F > 1:
            line["single_stream"] = {"frames_in_flight": 1, "ms_per_step": round(single["elapsed"] / args.steps * 1e3, 4),
                                     "value": round(rays * args.steps / single["elapsed"] / 1e6, 1), "unit": "Mray/s"}
        if other is not None:
            line["other_mode"] = other
        if world == 1 and not args.no_cpu:
            base, cpu_out = cpu_baseline(spheres, cam)
            line["cpu_baseline"] = base
            if args.mode == "parity":
                import numpy as np
                line["config"]["matches_cpu_oracle_bit_exact"] = bool(np.array_equal(last_pixels, cpu_out["pixels"]))
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
