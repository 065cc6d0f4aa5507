#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

A step = ONE FRAME of BASELINE.json configs[1]: Demo scene (6 spheres), 1920x1080, 64 samples
per pixel (= 64 passes of the reference's Config::updateRendering()), from the default seed
stream, rendered by the HIP path through the C ABI.  Inputs (seeds, scene tables, camera) are
resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode parity|fast] [--no-cpu] [--workload c2|c16|c3|c4|c5|...|scn:<scene>]

N > 1: one process per GPU (torch.distributed over RCCL); the image is sharded by interleaved 8-row
tiles and each frame ends with one gather of the packed pixels to rank 0 and the library's
de-interleave kernel there.  Started as the driver starts it (torch.distributed.run, WORLD_SIZE set)
this file is one rank; started plainly (`python bench.py --gpus 8`) it launches the N ranks itself, as
child processes, before anything touches a GPU.  Total work is fixed, so scaling is "strong".

At N > 1 the line also carries a `c4` block: BASELINE configs[3] (Demo, 3840x2160, 256 spp -- the configuration built for the 1/2/4/8 curve), a few
frames one at a time, every gathered frame compared with the unsharded one; and at any N `config.predicted_from_shards`: every shard of a 2 / 4 / 8-way
split of the workload (and of C4) rendered ALONE on rank 0's GPU (tools/shard_prediction.py) -- what each GPU of an N-GPU run will spend on the frame.
Other blocks of the N = 1 line: `host_inclusive` (the one-shot rt_render() call with the frame read back, wall clock), `first_frame`, `unseen_passes`,
`north_star_target` (16 spheres), `large_scene` (C3), `other_mode`, `cpu_baseline` (the reference's kernel as host C++ on the host cores).

Prints ONE JSON line (rank 0).  The headline -- `value`, `ms_per_step` -- is ONE FRAME AT A TIME: frame
k+1 starts when frame k is complete (at N > 1: gathered and assembled on rank 0), which is the
"ms/frame" BASELINE.json's metric names and the regime the `roofline` object and the rocprof summaries
describe (kernel time <= step time).  `frames_in_flight` beside it is the throughput of the same K
frames with F of them in flight on separate streams (a renderer serving several views).  Every gathered
frame of every region is compared with the unsharded frame on the device.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Performance only, and only for the frames-in-flight figure: HIP gives a process 4 hardware queues and lets
# further streams share them; two frames on one queue do not overlap at all.  Results never depend on it
# (tools/gather_stress.py runs the N > 1 frame loop clean without it; DESIGN.md section 3 has the story of the
# round-1 failure this variable used to paper over).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

W, H, SPP = 1920, 1080, 64            # the headline workload (BASELINE.json configs[1]); --workload changes them
TILE_ROWS = 8
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters: 64 FLOP/clk/SIMD, i.e. every instruction a fused multiply-add
FP32_UNFUSED_PEAK_TFLOPS = 78.65     # the same issue rate with one FLOP per instruction: what parity mode (no contraction allowed) can reach at most
HBM_PEAK_GBS = 8000.0                # same table
FLOP_PER_SPHERE_TEST = 20            # ray-sphere test
FLOP_PER_BOX_TEST = 12               # the hierarchy's slab test: six fused multiply-adds (one per box plane)
BYTES_PER_PIXEL_PER_LAUNCH = 32      # seeds 8 R + 8 W, colour 12 W, pixel 4 W
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def pmc_record(workload, mode):
    """Counter-derived figures of the committed rocprofv3 profile of this workload's kernel (profiles/pmc_traffic.json,
    written by tools/summarize_profile.py from separate --pmc passes of this file's own command): HBM bytes per launch,
    VALU instructions, active lanes.  Constants of a committed profile, labelled with their source -- not measured in this
    run.  None where no profile of the workload is committed."""
    try:
        data = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None
    return (data.get(workload) or {}).get(mode)


def fetch_calibration():
    """FETCH_SIZE / WRITE_SIZE over bytes for the render kernels' access shapes (profiles/r06_fetch_size_calibration.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r06_fetch_size_calibration.json")))["factors"]
    except (OSError, ValueError, KeyError):
        return None


def nan_scene(count):
    """The Demo scene followed by records with a NaN centre (they never hit; the reference's loop tests them all the same)."""
    import numpy as np
    from raytracing_simple_amd import api, host
    sph = np.zeros(count, api.SPHERE_DT)
    sph[:6] = host.demo_scene()
    sph["rad"][6:] = 1.0
    sph["p"][6:] = np.float32("nan")
    sph["c"][6:] = 0.5
    return sph, host.DEMO_ORIG, host.DEMO_TARGET


def two_classes(n_small, n_large):
    from tools import always_list_probe
    return always_list_probe.two_classes(n_small, n_large)


def library_build_id():
    """rt_build_id() of the product library this run renders with (a hash of csrc/, the public headers and the compiler flags)."""
    try:
        from raytracing_simple_amd import api
        return api.build_id()
    except Exception:       # noqa: BLE001 -- a library that cannot be asked has no identity to match
        return None


def walk_census(api, spheres, cam, w, h, spp):
    """What the hierarchy walk EXECUTES for one frame of this scene: the census instance of the diagnostics library
    (rt_trace_parity_pairs_census: the shipped walk + per-lane counters) renders the frame once, outside every timed
    region -- pair steps (two box tests each), leaf steps (8 sphere tests each), sphere tests of the always-list sweeps."""
    import ctypes as C
    with api.RtContext(w, h, diag=True) as c:
        c._check(c._lib.rt_debug_set_walk(c._h, 0, 0, 1))
        c.set_scene(spheres)
        c.set_camera(cam)
        c.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
        c.render_pass(spp, copy=False)
        raw = (C.c_ulonglong * 32)()
        c._check(c._lib.rt_debug_counters_raw(c._h, raw))
        st = c.stats()
    return {"box_tests": 2 * int(raw[21]), "leaf_sphere_tests": 8 * int(raw[23]), "always_sphere_tests": int(raw[29]),
            "pair_steps": int(raw[21]), "leaf_steps": int(raw[23]), "lanes_per_pair_step": round(raw[21] / max(raw[20], 1), 1),
            "lanes_per_leaf_step": round(raw[23] / max(raw[22], 1), 1), "reference_equivalent_sphere_tests": st["sphere_tests"]}


def scaling_bound(api, mode, spheres, cam, w, h, spp, frame_ms=None, device=0):
    """The perfect-balance estimate of ONE frame on N GPUs, as numbers: a pixel's samples consume one random
    stream in order (.cl:143-169), so however many GPUs share the image a frame lasts as long as its slowest wavefront.  That time
    is read here from what every launch leaves -- the wall clock of each tile's slowest wavefront, from its workgroup's start (table
    staging and barrier included), of an untimed, unsharded frame on THIS GPU AT FULL OCCUPANCY: the wavefront shared its SIMD
    with five or six others of a VALU-bound kernel, so on a GPU that holds fewer tiles than wave slots (8 GPUs at 1080p) it is
    shorter.  It is therefore NOT a floor and the ratio below NOT a ceiling: `predicted_from_shards` (tools/shard_prediction.py:
    every shard of the N-way split rendered alone on this GPU) is the measured figure the first SCALE record is to be read against."""
    import ctypes as C
    import numpy as np
    with api.RtContext(w, h, device=device, diag=True) as c:
        c.set_scene(spheres)
        c.set_camera(cam)
        c.set_mode(mode)
        for _ in range(2):                      # (the second frame walks its tiles heavy first, as the timed ones do)
            c.reset()
            c.render_pass(spp, copy=False)
        one_gpu_ms = c.stats()["last_kernel_ms"]
        cap = ((w + 7) // 8) * ((h + 7) // 8)
        cost = np.zeros(cap, np.uint32)
        n_tiles, valid = C.c_uint32(), C.c_int()
        c._check(c._lib.rt_debug_read_tile_order(c._h, None, cost.ctypes.data_as(C.c_void_p), cap, C.byref(n_tiles), C.byref(valid)))
        slowest_ms = float(cost[:n_tiles.value].max()) * 1e-5           # s_memrealtime ticks of 10 ns
    frame = frame_ms if frame_ms else one_gpu_ms
    return {"slowest_wavefront_under_full_occupancy_ms": round(slowest_ms, 4), "one_gpu_frame_ms": round(frame, 4),
            "perfect_balance_speedup": {str(n): round(frame / max(slowest_ms, frame / n), 2) for n in (2, 4, 8)},
            "perfect_balance_ms_per_frame": {str(n): round(max(slowest_ms, frame / n), 4) for n in (2, 4, 8)},
            "note": "an estimate, neither floor nor ceiling: the slowest wavefront was timed while it shared its SIMD with 5-6 others (a GPU holding "
                    "1/N of the tiles runs it faster), shards are taken as equally heavy, kernel time only (gather of the packed rows and launch "
                    "latency come on top); `predicted_from_shards` holds every shard measured alone; frames_in_flight is the figure that scales with N"}


def first_frame(api, mode, spheres, cam, w, h, spp, steady_ms):
    """A NEW scene's first frame -- what rt_render(scene, cam, out, w, h, spp), the call north_star names, is by definition: a
    fresh context (GPU warm, outside every timed region), rt_set_scene + rt_set_camera + ONE blocking frame.  `ms` is the device
    time between the events around everything that frame launched (pricing launches, probes if the scene is measured),
    `wall_ms` the host time of set_scene + the frame (tree build on the host, uploads), `vs_steady` = ms over the steady frame."""
    with api.RtContext(w, h) as c:
        c.set_mode(mode)
        c.set_camera(cam)
        t0 = time.perf_counter()
        c.set_scene(spheres)
        c.render_pass(spp, copy=False)
        wall = time.perf_counter() - t0
        st, ch = c.stats(), c.scene_choice()
        out = {"ms": round(st["last_kernel_ms"], 4), "launches": int(st["launches"]), "vs_steady": round(st["last_kernel_ms"] / steady_ms, 3),
               "wall_ms_with_set_scene": round(wall * 1e3, 3), "kernel": c.last_kernel}
        if ch["picked"] is not None:
            out["choice"] = ch["picked"] + (" (measured: four probe launches inside this frame)" if ch["hierarchy_ms_per_pass"] > 0
                                            else " (from the uploaded tree's surface areas: nothing measured)")
        # the same new scene on a context that is ALIVE -- it has just rendered another scene (what a host that keeps its context, or
        # calls rt_render again at the same size, pays for a new scene): tile costs and order dropped as above, buffers and queue warm
        from raytracing_simple_amd import host as _host, scenes as _scenes
        other = _host.demo_scene() if len(spheres) != 6 else _scenes.demo_plus(16)[0]
        live = []
        for _ in range(3):
            c.set_scene(other)
            c.reset()
            c.render_pass(spp, copy=False)
            c.set_scene(spheres)
            c.reset()
            c.render_pass(spp, copy=False)
            live.append(c.stats()["last_kernel_ms"])
        out["ms_on_a_live_context"] = round(sorted(live)[1], 4)
        out["live_vs_steady"] = round(out["ms_on_a_live_context"] / steady_ms, 3)
    return out


def host_inclusive(spheres, cam, w, h, spp, kernel_ms):
    """What the call north_star NAMES costs a host, beside the device-time headline: rt_render(scene, cam, out, w, h, spp) one-shot --
    asynchronous reset, scene (nothing uploaded when the records equal the previous call's), camera, the launch and the D2H of
    the frame (4 * w * h bytes through page-locked staging) -- as wall clock around the call.  Two sources: this process
    (its library is warm: 9 calls, the first of them apart, median of the other 8) and a FRESH process (the native host
    tools/rt_bench.cpp --oneshot 9: its first call pays HIP start-up, dlopen and the library's first context)."""
    import statistics
    from raytracing_simple_amd import api
    wall = []
    for _ in range(9):
        t0 = time.perf_counter()
        api.render(spheres, cam, w, h, spp)
        wall.append((time.perf_counter() - t0) * 1e3)
    out = {"call": "rt_render(scene, cam, out, w, h, spp): blocking, result in a host buffer (include/rt_api.h)",
           "wall_ms_median_of_8": round(statistics.median(wall[1:]), 4), "first_call_in_this_process_ms": round(wall[0], 3),
           "calls_ms": [round(v, 3) for v in wall], "d2h_bytes": 4 * w * h,
           "over_kernel_ms": round(statistics.median(wall[1:]) / kernel_ms, 3),
           "note": "`value` / `ms_per_step` above are device time with inputs and outputs resident in HBM (the contract of this line); this block is the same frame "
                   "through the one-shot host call, PCIe read-back included -- never `value`"}
    exe = os.path.join(ROOT, "raytracing_simple_amd", "rt_bench")
    if len(spheres) == 6 and os.path.exists(exe):         # (the native host renders the built-in Demo scene: the headline workload)
        try:
            res = subprocess.run([exe, "2", "1", "0", "--w", str(w), "--h", str(h), "--spp", str(spp), "--oneshot", "9"], capture_output=True, text=True, timeout=120)
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            ms = json.loads(lines[-1])["rt_render_wall_ms"]
            out["fresh_process"] = {"host": "raytracing_simple_amd/rt_bench (tools/rt_bench.cpp: the reference's Main.cpp without the window) --oneshot 9",
                                    "first_call_ms": round(ms[0], 3), "wall_ms_median_of_8": round(statistics.median(ms[1:]), 4)}
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError, IndexError) as e:
            out["fresh_process"] = {"error": repr(e)[:200]}
    return out


NORTH_STAR_PSNR_GATE_DB = 50.0       # BASELINE.json north_star: "PSNR >= 50 dB against it for multi-spp float accumulation"


def fast_mode_check(api, host, ctx, spp, parity_pixels):
    """RT_MODE_FAST on the workload `ctx` holds, against north_star's tolerance: one warm and one timed fast frame of the same
    sample count, its PSNR against the parity frame (which is bit-equal to the reference CPU path).  Outside every timed
    region; the context is handed back in parity mode.  No fast-mode rate may be quoted for a workload that fails the gate."""
    ctx.set_pixel_buffer(0, 0)
    ctx.set_mode(api.RT_MODE_FAST)
    for _ in range(2):
        ctx.reset()
        px = ctx.render_pass(spp)
    ms, kern = ctx.stats()["last_kernel_ms"], ctx.last_kernel
    ctx.set_mode(api.RT_MODE_PARITY)
    db = host.psnr(px, parity_pixels)
    return {"kernel": kern, "kernel_ms": round(ms, 4), "psnr_db_vs_parity": round(db, 2), "gate_db": NORTH_STAR_PSNR_GATE_DB,
            "meets_north_star_gate": bool(db >= NORTH_STAR_PSNR_GATE_DB)}


def roofline_block(kernel, kernel_ms, sphere_tests, n_pixels, n_spheres, workload, mode, census=None, choice=None):
    """The `roofline` object of one kernel: algorithmic work per launch over its measured duration against the FP32 vector
    peak, with the HBM view nested beside it.  Plain sweeps execute exactly the reference's tests, so there the algorithmic
    figure (20 FLOP x counted sphere tests) is the executed one.  The hierarchy executes far fewer: its `achieved` / `frac`
    come from what the walk executes (12 FLOP per box test, 20 per sphere test: `executed_work`, counted by the census
    instance), and the reference-equivalent rate stands beside it as what it is -- a rate, not a fraction of a peak."""
    sec = kernel_ms * 1e-3
    ref_flops = FLOP_PER_SPHERE_TEST * sphere_tests
    alg_bytes = BYTES_PER_PIXEL_PER_LAUNCH * n_pixels + 44 * n_spheres + 60
    out = {"bound": "valu-fp32", "kernel": kernel, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "kernel_ms": round(kernel_ms, 4)}
    if mode == "parity":
        out["peak_unfused"] = FP32_UNFUSED_PEAK_TFLOPS
    if census is not None:
        flops = FLOP_PER_BOX_TEST * census["box_tests"] + FLOP_PER_SPHERE_TEST * (census["leaf_sphere_tests"] + census["always_sphere_tests"])
        out.update({"achieved": round(flops / sec / 1e12, 3), "frac": round(flops / sec / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5),
                    "algorithmic_flops_per_launch": flops,
                    "work_model": "EXECUTED work of the hierarchy walk: 12 FLOP x box tests (6 fused multiply-adds each) + 20 FLOP x sphere tests "
                                  "(leaf visits and always-list sweeps), counted on this frame by the census instance rt_trace_parity_pairs_census",
                    "executed_work": census,
                    "reference_equivalent": {"sphere_tests": sphere_tests, "flops": ref_flops, "TFLOP_s": round(ref_flops / sec / 1e12, 2),
                                             "note": "what the reference's sweep (every ray against every sphere in scene order, up to the first blocker for "
                                                     "shadow rays) would execute for the same rays -- rt_stats.sphere_tests, equal to the oracle's; the walk "
                                                     "reaches the same answers without executing them, so this is a rate, not a fraction of the peak"}})
    elif "_pairs" in kernel:
        # a hierarchy walk whose tables the census instance cannot hold (it stages everything in LDS: beyond about 6 000 spheres it has no room):
        # what the walk executes is not counted here, so no fraction of a peak is claimed -- the reference-equivalent rate stands alone
        out.update({"achieved": None, "frac": None, "algorithmic_flops_per_launch": None,
                    "work_model": "hierarchy walk, executed work NOT counted for this scene (the census instance stages the whole tables in LDS and this scene's do "
                                  "not fit): no fraction of the peak is claimed; DESIGN.md section 5.3 holds this kernel's counters and its bound",
                    "reference_equivalent": {"sphere_tests": sphere_tests, "flops": ref_flops, "TFLOP_s": round(ref_flops / sec / 1e12, 2),
                                             "note": "what the reference's sweep would execute for the same rays (rt_stats.sphere_tests, equal to the oracle's): a rate, "
                                                     "not a fraction of the peak"}})
    else:
        out.update({"achieved": round(ref_flops / sec / 1e12, 3), "frac": round(ref_flops / sec / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5),
                    "algorithmic_flops_per_launch": ref_flops,
                    "work_model": "the reference's sweep, which this kernel executes test for test: every ray tests the spheres in scene order (all of them, "
                                  "or up to its first blocker): 20 FLOP x rt_stats.sphere_tests"})
    if mode == "parity" and out["achieved"] is not None:        # parity mode may not fuse a multiply with an add: its ceiling is the issue rate at one FLOP per instruction
        out["frac_unfused"] = round(out["achieved"] / FP32_UNFUSED_PEAK_TFLOPS, 5)
    if choice is not None and choice.get("picked") is not None and choice["hierarchy_ms_per_pass"] > 0:
        out["measured_choice"] = {"picked": choice["picked"], "hierarchy_ms_per_pass": round(choice["hierarchy_ms_per_pass"], 4),
                                  "sweep_ms_per_pass": round(choice["sweep_ms_per_pass"], 4),
                                  "note": "the library timed each form on this scene, warm, in the same tile order (rt_scene_choice); the sweep is the "
                                          "wave-ballot any-hit instance (rt_trace_*_coop)"}
    elif choice is not None and choice.get("picked") is not None:
        out["estimated_choice"] = {"picked": choice["picked"],
                                   "note": "settled at rt_set_scene from the surface areas of the tree the host built (predicted walk / sweep time per ray outside "
                                           "0.75 .. 1.33): nothing was measured, the scene's first frame cost what a frame costs (first_frame)"}
    out["hbm"] = {"algorithmic_bytes_per_launch": alg_bytes, "achieved": round(alg_bytes / sec / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 6)}
    pm = pmc_record(workload, mode)
    out["traffic"] = None
    lib_id = library_build_id()
    if pm and (pm.get("kernel", kernel) != kernel or pm.get("build_id") != lib_id):
        # counters of a committed profile describe the code they were measured on, nothing else
        out["counters_withheld"] = ("profiles/pmc_traffic.json holds counters of kernel %s from the library with build id %s; this run is kernel %s of "
                                    "library %s: `traffic` and `executed` are not printed beside code they were not measured on (tools/profile_gpu.sh + "
                                    "tools/summarize_profile.py make a record for this library)" % (pm.get("kernel"), pm.get("build_id"), kernel, lib_id))
        pm = None
    if pm:
        # FETCH_SIZE on gfx950 counts HALF the bytes of reads that arrive as wide coalesced segments (MI355X_MICROARCH.md, HBM section).  Which of
        # this library's access shapes that applies to is MEASURED, not assumed (tools/ubench/fetch_size.hip, profiles/r06_fetch_size_calibration.json:
        # known byte counts, 1 GiB streamed between launches): a single-wavefront workgroup's 8x8 tile reading its seed pairs in 64-byte row
        # segments (the *_w1 instances) is counted in full (FETCH_SIZE / bytes = 1.0006), the same read by 256-thread workgroups -- 256 contiguous
        # bytes per row, every other instance -- at exactly half (0.5006), stores exactly (1.0).  Both raw sums are printed on every instance;
        # `traffic` is WRITE_SIZE + FETCH_SIZE / (the factor measured for the instance's workgroup shape).
        raw, fixed = pm.get("hbm_bytes_per_launch"), pm.get("hbm_bytes_per_launch_fetch_x2")
        cal = fetch_calibration()
        wide = not kernel.endswith("_w1")
        out["traffic_raw"], out["traffic_fetch_x2"] = raw, fixed
        if cal and pm.get("fetch_raw_bytes") is not None and pm.get("write_bytes") is not None:
            k = cal["fetch_over_bytes_tile32_uint2_256B_rows" if wide else "fetch_over_bytes_tile8_uint2_64B_row_segments"]
            out["traffic"] = int(pm["write_bytes"] / cal["write_over_bytes_frame_epilogue"] + pm["fetch_raw_bytes"] / k)
            out["traffic_basis"] = ("WRITE_SIZE / %.4f + FETCH_SIZE / %.4f: the factors FETCH_SIZE / WRITE_SIZE over bytes measured for this instance's workgroup shape (%s) "
                                    "on a known byte count -- profiles/r06_fetch_size_calibration.json" % (cal["write_over_bytes_frame_epilogue"], k,
                                    "256-thread workgroups, 256-byte rows" if wide else "one wavefront per 8x8 tile, 64-byte row segments"))
        else:
            out["traffic"] = fixed if (wide and fixed) else raw
            out["traffic_basis"] = "WRITE_SIZE + 2 x FETCH_SIZE (256-byte rows)" if (wide and fixed) else "WRITE_SIZE + FETCH_SIZE as counted (64-byte row segments)"
        if out["traffic"] and out["traffic"] < alg_bytes:
            out["traffic_note"] = ("below the algorithmic bytes by %.1f %%: the frame loop renders the same frame again and part of the seed stream it reads is still in the 32 MiB of L2 "
                                   "from the frame before (the calibration's cold launches count 1.0006 of the bytes) -- read `hbm.algorithmic_bytes_per_launch` as the floor"
                                   % (100.0 * (1.0 - out["traffic"] / alg_bytes)))
        if out["traffic"] and out["traffic"] > 1.15 * alg_bytes:
            out["traffic_note"] = ("above the algorithmic bytes: a wavefront's 8x8 square stores its colours (12 B per pixel) and seeds in 32-96 byte segments, "
                                   "not whole lines; the kernel is VALU-bound at under 1 % of the HBM peak")
        if pm.get("valu_insts_per_launch"):
            # what the VALU actually issues (PMC of the committed profile, same command): the time its instructions
            # need at full issue rate, and how much of this run's kernel time that is
            floor_ms = pm["valu_busy_frac_single_stream"] * pm["profiled_kernel_ms"]
            out["executed"] = {"valu_insts_per_launch": pm["valu_insts_per_launch"], "active_lane_frac": pm["active_lane_frac"],
                               "valu_issue_floor_ms": round(floor_ms, 4), "valu_busy_frac": round(floor_ms / kernel_ms, 4),
                               "l2_hit_rate_whole_kernel": pm.get("l2_hit_rate"), "l2_hit_rate_staged_tables": pm.get("l2_hit_rate_staged_tables"),
                               "l2_hit_rate_note": "whole kernel = every TCC request of the launch (mostly the once-per-pixel seed / colour streams, which miss by design); "
                                                   "staged tables = the workgroups' reads of the sphere tables into LDS in isolation (north_star's figure; "
                                                   "rt_debug_stage_tables under --pmc, tools/pmc_staging.sh, same library)",
                               "lds_bank_conflict_frac": pm.get("lds_bank_conflict_frac"),
                               "source": pm.get("source", "profiles/pmc_traffic.json")}
    return out


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


def cpu_baseline(spheres, cam, w, h, spp, reference_too=True):
    """The reference's CPU path on the host cores, same workload, same run: the reference's own
    kernel compiled as host C++ (oracle/_ref, kind "reference") when that build travelled with the
    snapshot, and the oracle (the CPU restatement, kind "port") -- the headline is the reference
    build when it is there.  Both are checkers: neither is part of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = host_cores()
    t0 = time.time()
    out = O.render(spheres, cam, w, h, spp, threads=cores)
    dt = time.time() - t0
    st = out["stats"]
    rays = st["samples"] + st["shadow_calls"]
    where = f"on {cores} threads (host shows {os.cpu_count()} CPUs; cgroup quota / affinity grant {cores})"
    port = {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"full workload: {w}x{h} x {spp} spp, {rays} rays in {dt:.2f} s "
                      f"({st['samples'] / dt / 1e6:.1f} Msample/s) {where}",
            "ms_per_frame": round(dt * 1e3, 1)}
    if not (reference_too and O.ref_available()):
        return port, out
    import numpy as np
    t0 = time.time()
    ref = O.ref_render_mt(spheres, cam, w, h, spp, cores)
    dt = time.time() - t0
    base = {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "reference",
            "sample": f"the reference's kernel source compiled as host C++ (oracle/_ref), full workload: {w}x{h} x {spp} "
                      f"passes, {rays} rays in {dt:.2f} s {where}",
            "ms_per_frame": round(dt * 1e3, 1),
            "equals_port_bit_exact": bool(np.array_equal(ref["pixels"], out["pixels"])),
            "port": {k: port[k] for k in ("value", "ms_per_frame", "kind")}}
    return base, out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes BEFORE this
    process touches a GPU, pass their output through, leave with their exit code."""
    import socket
    import torch                      # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1" and args.gpus > 3 and os.environ.get("RT_REHEARSAL_OVERSUBSCRIBE") != "1":
        # the platform precondition of DESIGN.md section 3: more processes' hardware queues on ONE GPU than its scheduler keeps
        # resident (four ranks of this bench) and a dispatch can lose its writes -- a wrong frame there is not a library bug
        raise SystemExit(f"bench.py --gpus {args.gpus} with RT_BENCH_SINGLE_DEVICE=1: at most 3 ranks are rehearsed on one device (queue "
                         "oversubscription loses writes on this driver stack: DESIGN.md section 3, profiles/r02_stale_seed/); "
                         "RT_REHEARSAL_OVERSUBSCRIBE=1 overrides, for studying exactly that")
    if have < args.gpus and os.environ.get("RT_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} HIP device(s); nothing is measured on fewer "
                         "GPUs than asked for (RT_BENCH_SINGLE_DEVICE=1 rehearses the ranks on one device, with gloo)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def inproc_child(args):
    """The in-library multi-device path (rt_create_multi: one process, one stream per device, ncclSend/ncclRecv
    gather inside librt_hip.so, de-interleave kernel) on the same workload: K frames one at a time
    through rt_render_pass, the last one compared with a one-device render.  Run by rank 0 as a child process
    after the ranks' own measurement; prints one JSON object."""
    import numpy as np
    from raytracing_simple_amd import api, host
    n = args.inproc_child
    spheres, cam = host.demo_scene(), host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W, H)
    mode = api.RT_MODE_FAST if args.mode == "fast" else api.RT_MODE_PARITY
    with api.RtContext(W, H) as one:
        one.set_scene(spheres); one.set_camera(cam); one.set_mode(mode)
        want = one.render_pass(SPP)
    devices = list(range(n)) if os.environ.get("RT_BENCH_SINGLE_DEVICE") != "1" else [0] * n
    with api.RtContext(W, H, devices=devices, tile_rows=TILE_ROWS) as ctx:
        ctx.set_scene(spheres); ctx.set_camera(cam); ctx.set_mode(mode)
        for _ in range(max(args.warmup, 2)):
            ctx.reset_async(); ctx.render_pass(SPP, copy=False)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.reset_async(); ctx.render_pass(SPP, copy=False)         # frame assembled on device 0 when this returns
        dt = time.perf_counter() - t0
        got = ctx.read_pixels()                                         # the last frame, for the comparison below
        st = ctx.stats()
    rays = st["samples"] + st["shadow_rays"]
    print(json.dumps({"n_gpus": n, "path": "rt_create_multi: one process, in-library gather"
                      + (" (one-GPU rehearsal: D2D copies stand in for ncclSend/ncclRecv)" if len(set(devices)) < n
                         else " (ncclSend/ncclRecv across distinct devices: before round 3's driver run this branch had executed nowhere -- "
                              "frame_equals_single_device is its check)"),
                      "steps": args.steps, "ms_per_frame": round(dt / args.steps * 1e3, 4), "value": round(rays * args.steps / dt / 1e6, 1),
                      "unit": "Mray/s",
                      "slowest_shard_kernel_ms": round(st["last_kernel_ms"], 4),
                      "frame_equals_single_device": bool(np.array_equal(got, want))}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["parity", "fast"], default="parity")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--workload", default="c2",
                    help="c2 = the headline (default; what the driver measures); c16 / c3 / c4 / c5 = the other BASELINE configurations; box120 / r2048 / r8192 / "
                         "r65536 / r262144 / nan9800 / nan9800hd / dust10k = scenes that run on the shipped instances no BASELINE configuration reaches (rt_trace_*_coop, _pairs_m, _pairs_g, "
                         "rt_trace_*_g), for their profiles; scn:<scene> = one of the reference's own scenes (demo, simple, cornell, cornell_large, caustic, "
                         "caustic3, demo_scn, complex, cornell_test, complex_test: sphere array and camera from tests/golden/) at the reference's 800x600, 64 spp")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames kept in flight per rank in the throughput region (0 = 2 for N<=2, 3 for N<=4, 6 beyond)")
    ap.add_argument("--no-extras", action="store_true", help="headline regions only (profiling runs)")
    ap.add_argument("--inproc-child", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.inproc_child:
        return inproc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    # ONE line on stdout: whatever the libraries underneath print there (RCCL's version banner at communicator
    # creation, for one) goes to stderr instead; the JSON line is written to the real stdout at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch                       # plumbing: streams, events, torch.distributed (RCCL)
    import torch.distributed as dist

    from raytracing_simple_amd import api, host, scenes
    from raytracing_simple_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # rehearsal knobs for a one-GPU box (never set by the driver): every rank on device 0, gloo instead of RCCL
    if os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl" if os.environ.get("RT_BENCH_SINGLE_DEVICE") != "1" else "gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    global W, H, SPP
    workloads = {
        "c2": ("C2: Demo scene (6 spheres)", lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 1920, 1080, 64),
        "c16": ("north-star target scene: Demo + 10 spheres (16)", lambda: scenes.demo_plus(16), 1920, 1080, 64),
        "c3": ("C3: 1024 random spheres", lambda: scenes.random_spheres(1024), 1920, 1080, 16),
        "c4": ("C4: Demo scene (6 spheres)", lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 3840, 2160, 256),
        "c5": ("C5: 64-sphere mirror box, depth 8", lambda: scenes.mirror_box(64), 1920, 1080, 64),
        "box120": ("closed box of 120 mirror / glass spheres (the 4-wavefront cooperative sweep)", lambda: scenes.mirror_box(120), 1920, 1080, 8),
        "r2048": ("2048 random spheres (hierarchy pairs in LDS, slots read from HBM / L2)", lambda: scenes.random_spheres(2048), 1920, 1080, 8),
        "r8192": ("8192 random spheres (hierarchy read from HBM / L2)", lambda: scenes.random_spheres(8192), 1920, 1080, 4),
        "r65536": ("65536 random spheres (hierarchy read from HBM / L2)", lambda: scenes.random_spheres(65536), 1920, 1080, 4),
        "r262144": ("262144 random spheres = RT_MAX_SPHERES (hierarchy read from HBM / L2)", lambda: scenes.random_spheres(262144), 1920, 1080, 4),
        "nan9800": ("the Demo scene + 9794 records whose centre is not a number: thousands of records of which fewer than 56 are finite small spheres -- no hierarchy, "
                    "and a table beyond the sweep's LDS budget: the plain sweep over a table in HBM / L2 through the scalar cache (rt_trace_*_g), the fallback that keeps every input renderable", lambda: nan_scene(9800), 640, 360, 1),
        "dust10k": ("6 000 small spheres (radius 0.02-0.05) among 4 000 objects fifty times their size, a ground sphere, a light: two size classes, both in the hierarchy since "
                    "round 6 (tools/always_list_probe.py; run with --no-cpu: the CPU check of 10 002 spheres at this size takes an hour)", lambda: two_classes(6000, 4000), 1920, 1080, 4),
        "nan9800hd": ("the nan9800 scene at 1920x1080: 32 400 wavefronts instead of 3 600 (640x360 leaves the GPU's 6 144 wavefront slots under-filled; the small size is the "
                      "one whose CPU check takes seconds -- run this one with --no-cpu)", lambda: nan_scene(9800), 1920, 1080, 1),
    }
    if args.workload.startswith("scn:"):
        from tools import reference_scenes
        scn = args.workload[4:]
        if scn not in reference_scenes.FIXTURES:
            raise SystemExit(f"--workload {args.workload}: scenes are {sorted(reference_scenes.FIXTURES)}")
        workloads[args.workload] = (f"the reference's {reference_scenes.SOURCE.get(scn, 'Scene/' + scn + '.scn')} as its loader hands it to the kernel, the reference's native window",
                                    lambda: reference_scenes.load_scene(scn), reference_scenes.W, reference_scenes.H, reference_scenes.SPP)
    if args.workload not in workloads:
        raise SystemExit(f"--workload {args.workload}: one of {sorted(workloads)} or scn:<scene>")
    wl_name, wl_maker, W, H, SPP = workloads[args.workload]
    spheres, cam_orig, cam_target = wl_maker()
    cam = host.compute_camera(cam_orig, cam_target, W, H)
    mode = api.RT_MODE_FAST if args.mode == "fast" else api.RT_MODE_PARITY

    # F contexts, one HIP stream each (the contexts' own: created back to back they land on distinct hardware
    # queues, which streams from torch's pool did not always do)
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if world <= 2 else (3 if world <= 4 else 6))

    class FrameLoop:
        """The frame loop of one workload on this rank: `n_ctx` sharded contexts (one stream each), at N > 1 the gather buffers, the
        unsharded frame every gathered frame is compared with (rank 0) and the timed regions over them."""

        def __init__(self, sph, camera, w, h, spp, n_ctx):
            self.w, self.h, self.spp, self.n_ctx = w, h, spp, n_ctx
            self.ctxs = []
            for _ in range(n_ctx):
                c = api.RtContext(w, h, device=local_rank, rank=rank, nranks=world, tile_rows=TILE_ROWS)
                c.set_scene(sph)
                c.set_camera(camera)
                c.set_mode(mode)
                self.ctxs.append(c)
            # The library measures per scene what it cannot know: hierarchy against sweep on a large scene's first launches (a blocking call of
            # 16 passes or more holds them all), cooperative any-hit against plain on a scene of 4-11 spheres' second and third whole frames
            # (rt_launch.hip launch_small).  Four untimed blocking frames per context settle both before any timed region, whatever --warmup says
            for c in self.ctxs:
                for _ in range(4):
                    c.reset_async()
                    c.render_pass(max(spp, 16), copy=False)
            self.streams = [torch.cuda.ExternalStream(c.stream, device=dev) for c in self.ctxs]
            self.gather, self.want_dev, self.mismatch, self.whole_kernel_ms = None, None, None, None
            self.frame_no, self.frames_checked = 0, 0
            if world > 1:
                self.gather = rdist.FrameGatherer(h, w, rank, world, TILE_ROWS, dev, slots=2 * n_ctx)
                self.gather.pending_check = [False] * self.gather.slots
                if rank == 0:
                    # the frame every gathered frame must equal: an unsharded render on this GPU
                    with api.RtContext(w, h, device=local_rank) as whole:
                        whole.set_scene(sph); whole.set_camera(camera); whole.set_mode(mode)
                        whole.render_pass(spp, copy=False)
                        whole.reset()
                        self.want_dev = torch.from_numpy(whole.render_pass(spp).view(np.int32).reshape(h, w).copy()).to(dev)
                        self.whole_kernel_ms = whole.stats()["last_kernel_ms"]     # the unsharded frame on this GPU (its second rendering: tiles heavy first)
                    self.mismatch = torch.zeros((), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()        # the set-up ran on torch's stream; the contexts' streams are non-blocking
                self.gather.gather(0)           # plumbing, not a step: RCCL builds its communicator and point-to-point
                torch.cuda.synchronize()        # channels on first use (seconds); keep that out of every timed region
                dist.barrier()

        def collect(self, k):
            """Frame k's gather is complete and assembled (rank 0): compare it with the unsharded frame ON THE DEVICE,
            then poison the buffers so that a frame that is NOT written again would show."""
            g = self.gather
            full = g.wait(k)
            if rank == 0 and full is not None and g.pending_check[k % g.slots]:
                self.mismatch.add_((full != self.want_dev).sum())
                full.fill_(-1)
                g.pending_check[k % g.slots] = False
                self.frames_checked += 1

        def step(self, in_flight, ev=None):
            k = self.frame_no
            self.frame_no += 1
            c, st = self.ctxs[k % in_flight], self.streams[k % in_flight]
            with torch.cuda.stream(st):
                if self.gather is not None:
                    self.collect(k)                                 # slot free again (the frame that used it 2F frames ago)
                    buf = self.gather.local_slot(k)
                    buf.fill_(-1)                                   # poison: every pixel must come from THIS frame's launch
                    c.set_pixel_buffer(buf.data_ptr(), buf.numel())   # render straight into the send buffer
                c.reset_async(st.cuda_stream)
                if ev:
                    ev[0].record(st)
                c.render_async(self.spp, st.cuda_stream)
                if ev:
                    ev[1].record(st)
                if self.gather is not None:
                    self.gather.gather(k, async_op=True)            # queued behind the launch, not waited for
                    self.gather.pending_check[k % self.gather.slots] = True
                    if in_flight == 1:
                        self.collect(k)                             # one frame at a time: complete before the next one starts
                        st.synchronize()
            return c

        def drain(self):
            if self.gather is not None:
                for k in range(self.frame_no - 2 * self.n_ctx, self.frame_no):
                    if k >= 0:
                        with torch.cuda.stream(self.streams[0]):
                            self.collect(k)

        @staticmethod
        def sync():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def timed_region(self, in_flight, steps, warmup, with_events=True):
            """W untimed + exactly `steps` timed frames; barrier + synchronize on both sides.
            Per-launch HIP events only where asked: an event pair around every launch costs the
            overlapped region its overlap (measured), and a per-launch duration means little there."""
            for _ in range(warmup):
                self.step(in_flight)
            self.drain()
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                      for _ in range(steps)] if with_events else None
            self.sync()
            t0 = time.perf_counter()
            last = None
            for k in range(steps):
                last = self.step(in_flight, events[k] if events else None)
            self.drain()                                            # every frame gathered, assembled and checked
            self.sync()
            elapsed = time.perf_counter() - t0
            kernel_ms = (sum(a.elapsed_time(b) for a, b in events) / max(steps, 1)) if events else None
            return elapsed, kernel_ms, last

        def frame_counters(self, c=None):
            """Exact ray / test counts of one frame of this rank (the same for every frame): from one more,
            synchronous render after a synchronous reset, outside every timed region."""
            c = c or self.ctxs[0]
            c.set_pixel_buffer(0, 0)
            c.reset()
            c.render_pass(self.spp, copy=False)
            return c.stats()

        def close(self):
            for c in self.ctxs:
                c.close()
            self.ctxs = []

    def all_ranks(values, op):
        t = torch.tensor(values, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    loop = FrameLoop(spheres, cam, W, H, SPP, F)
    ctxs, side_streams, ctx = loop.ctxs, loop.streams, loop.ctxs[0]

    def timed_region(contexts, streams, in_flight, steps, warmup, with_events=True):
        """(the extras below time other scenes through loops of their own: a FrameLoop around contexts that exist already)"""
        lp = FrameLoop.__new__(FrameLoop)
        lp.w, lp.h, lp.spp, lp.n_ctx, lp.ctxs, lp.streams = W, H, SPP, len(contexts), contexts, streams
        lp.gather, lp.want_dev, lp.mismatch, lp.frame_no, lp.frames_checked = None, None, None, 0, 0
        return lp.timed_region(in_flight, steps, warmup, with_events)

    def frame_counters(c):
        return loop.frame_counters(c)

    # ---- the headline: one frame at a time -------------------------------------------------------------
    el1, kernel_ms, last1 = loop.timed_region(1, args.steps, args.warmup, with_events=True)
    last_pixels = last1.read_pixels() if world == 1 else None       # the last TIMED frame (checked against the oracle below)
    # ---- the same K frames with F in flight (throughput) ------------------------------------------------
    elF = None
    if F > 1 and not args.no_extras:
        elF, _, _ = loop.timed_region(F, args.steps, args.warmup, with_events=False)
    st = loop.frame_counters(ctx)
    kernel_name = ctx.last_kernel           # the instance the library renders this scene with (rt_last_kernel)
    choice = ctx.scene_choice()             # large scenes: what the library's own measurement of hierarchy against sweep said

    samples, closest, shadow, tests = (int(v) for v in all_ranks([st["samples"], st["closest_rays"], st["shadow_rays"], st["sphere_tests"]],
                                                                  dist.ReduceOp.SUM if world > 1 else None))
    el1_max, kernel_ms_max = all_ranks([el1, kernel_ms], dist.ReduceOp.MAX if world > 1 else None)
    elF_max = all_ranks([elF], dist.ReduceOp.MAX if world > 1 else None)[0] if elF is not None else None
    frames_ok = None
    if world > 1:
        bad = int(loop.mismatch.item()) if rank == 0 else 0
        frames_ok = {"frames_checked": loop.frames_checked, "wrong_pixels": bad} if rank == 0 else None

    # ---- extras, N = 1, outside the headline's timed regions -------------------------------------------
    other, target, in_library, large, unseen, first = None, None, None, None, None, None
    if world == 1 and not args.no_extras:
        # The headline renders the SAME frame K times (reset + SPP passes, fixed seed stream: what makes it checkable against the
        # oracle), and the library schedules a launch from what the launch before cost (tile order) -- costs that
        # are exact for a frame rendered again.  The same launch on passes it has NOT seen: SPP more passes of the running image
        # per launch, no reset in between (what a progressive renderer does all day).
        ctx.set_pixel_buffer(0, 0)
        ctx.reset()
        ctx.render_pass(SPP, copy=False)
        st_a = ctx.stats()
        ms_u = []
        for _ in range(6):
            ctx.render_pass(SPP, copy=False)
            ms_u.append(ctx.stats()["last_kernel_ms"])
        st_b = ctx.stats()
        rays_u = (st_b["samples"] + st_b["shadow_rays"]) - (st_a["samples"] + st_a["shadow_rays"])
        unseen = {"what": "launches of %d passes continuing the running image (passes %d .. %d): random numbers the costs behind the tile order and "
                          "the tile order has never seen" % (SPP, SPP, 7 * SPP - 1),
                  "launches": 6, "kernel_ms": round(sum(ms_u) / 6, 4), "value": round(rays_u / sum(ms_u) / 1e3, 1), "unit": "Mray/s (kernel time)",
                  "headline_kernel_ms": round(kernel_ms, 4)}
        first = {"what": "a new scene's first frame on a warm GPU: `ms` on a FRESH context (rt_create, rt_set_scene, rt_set_camera, one blocking frame: device ms "
                         "between the events around everything it launched), `ms_on_a_live_context` the same new scene on a context that has just rendered "
                         "another one; vs_steady / live_vs_steady = over this line's steady kernel time of the same frame",
                 "headline": first_frame(api, mode, spheres, cam, W, H, SPP, kernel_ms)}
        # the other arithmetic mode on the same workload
        other_mode = api.RT_MODE_FAST if mode == api.RT_MODE_PARITY else api.RT_MODE_PARITY
        for c in ctxs:
            c.set_mode(other_mode)
        el_o1, k_o1, last_o = timed_region(ctxs, side_streams, 1, 8, 2, with_events=True)
        px_o = last_o.read_pixels()
        el_oF, _, _ = timed_region(ctxs, side_streams, F, 8, 2, with_events=False)
        st_o = frame_counters(ctxs[0])
        rays_o = st_o["samples"] + st_o["shadow_rays"]
        other = {"mode": "fast" if other_mode == api.RT_MODE_FAST else "parity", "ms_per_step": round(el_o1 / 8 * 1e3, 4),
                 "value": round(rays_o * 8 / el_o1 / 1e6, 1), "unit": "Mray/s", "kernel_ms": round(k_o1, 4),
                 "frames_in_flight": {"frames": F, "ms_per_step": round(el_oF / 8 * 1e3, 4), "value": round(rays_o * 8 / el_oF / 1e6, 1)},
                 "psnr_db_vs_headline_mode": round(host.psnr(px_o, last_pixels), 2), "gate_db": NORTH_STAR_PSNR_GATE_DB,
                 "meets_north_star_gate": bool(host.psnr(px_o, last_pixels) >= NORTH_STAR_PSNR_GATE_DB),
                 "gate_note": "north_star's tolerance for the fused mode: PSNR >= 50 dB against the reference CPU path (= the parity frame, bit for bit) "
                              "at this workload's sample count; a fast-mode rate counts only where this is true (DESIGN.md section 3, tools/fast_gate.py)"}
        for c in ctxs:
            c.set_mode(mode)
        # the north-star target workload (16-sphere scene, 1080p x 64 spp, >= 10 Gray/s asked), driver-timed
        if args.workload == "c2":
            sph16, o16, t16 = scenes.demo_plus(16)
            cam16 = host.compute_camera(o16, t16, W, H)
            loop16 = FrameLoop(sph16, cam16, W, H, SPP, F)
            c16 = loop16.ctxs
            k16 = max(5, args.steps // 2)
            el16, kms16, last16 = loop16.timed_region(1, k16, 2, with_events=True)
            px16 = last16.read_pixels()
            el16F, _, _ = loop16.timed_region(F, k16, 2, with_events=False)
            st16 = loop16.frame_counters()
            rays16 = st16["samples"] + st16["shadow_rays"]
            target = {"workload": "north-star target: Demo + 10 spheres (16), 1920x1080, 64 spp, default seed stream",
                      "asked_Mray_s": 10000.0, "steps": k16, "ms_per_step": round(el16 / k16 * 1e3, 4),
                      "value": round(rays16 * k16 / el16 / 1e6, 1), "unit": "Mray/s", "kernel": c16[0].last_kernel, "kernel_ms": round(kms16, 4),
                      "roofline": roofline_block(c16[0].last_kernel, kms16, st16["sphere_tests"], W * H, len(sph16), "c16", args.mode),
                      "frames_in_flight": {"frames": F, "ms_per_step": round(el16F / k16 * 1e3, 4), "value": round(rays16 * k16 / el16F / 1e6, 1)}}
            first["north_star_target"] = first_frame(api, mode, sph16, cam16, W, H, SPP, kms16)
            if mode == api.RT_MODE_PARITY:
                target["fast_mode"] = fast_mode_check(api, host, c16[0], SPP, px16)
            if not args.no_cpu and mode == api.RT_MODE_PARITY:
                base16, cpu16 = cpu_baseline(sph16, cam16, W, H, SPP, reference_too=False)
                target["matches_cpu_oracle_bit_exact"] = bool(np.array_equal(px16, cpu16["pixels"]))
                target["cpu_port_ms_per_frame"] = base16["ms_per_frame"]
            loop16.close()
        # a large scene (BASELINE configs[2]: 1024 random spheres, 1080p x 16 spp), blocking calls: the library builds a
        # hierarchy for it, times it against the sweep on the first two frames and renders the rest with the faster form
        if args.workload == "c2":
            try:
                sph3, o3, t3 = scenes.random_spheres(1024)
                cam3 = host.compute_camera(o3, t3, W, H)
                with api.RtContext(W, H, device=local_rank) as cl:
                    cl.set_scene(sph3); cl.set_camera(cam3); cl.set_mode(mode)
                    for _ in range(3):
                        cl.reset_async(); cl.render_pass(16, copy=False)
                    k3 = max(5, args.steps // 2)
                    t0 = time.perf_counter()
                    for _ in range(k3):
                        cl.reset_async(); cl.render_pass(16, copy=False)
                    dt3 = time.perf_counter() - t0
                    st3, ch3, kern3 = cl.stats(), cl.scene_choice(), cl.last_kernel
                    fast3 = fast_mode_check(api, host, cl, 16, cl.read_pixels()) if mode == api.RT_MODE_PARITY else None
                    # the same scene MOVING: every frame rewrites the records on the device (rt_update_spheres_async, the whole range) and
                    # the hierarchy is rebuilt on the stream -- its shape chosen by surface area by the device's own build kernel
                    moved3 = api.as_spheres(sph3).copy()
                    upd_ms = []
                    for k in range(6):
                        moved3["p"][1:, 0] += np.float32(0.05)          # (every sphere but the ground drifts: an animation's frame-to-frame step)
                        cl.update_spheres(0, moved3)
                        cl.reset_async(); cl.render_pass(16, copy=False)
                        upd_ms.append(cl.stats()["last_kernel_ms"])
                    moving3 = {"what": "frames of the same scene with every sphere drifting 0.05 per frame: rt_update_spheres_async of every record, tree rebuilt on the stream by rt_bvh_build_sah_kernel, tile order sorted again from the frame before",
                               "kernel_ms": round(sorted(upd_ms[1:])[len(upd_ms[1:]) // 2], 4), "kernel": cl.last_kernel}
                first["large_scene"] = first_frame(api, mode, sph3, cam3, W, H, 16, st3["last_kernel_ms"])
                cen3 = walk_census(api, sph3, cam3, W, H, 16) if "_pairs" in kern3 else None
                large = {"workload": "C3: 1024 random spheres, 1920x1080, 16 spp, default seed stream", "steps": k3,
                         "ms_per_step": round(dt3 / k3 * 1e3, 4), "value": round((st3["samples"] + st3["shadow_rays"]) * k3 / dt3 / 1e6, 1),
                         "unit": "Mray/s", "kernel": kern3, "kernel_ms": round(st3["last_kernel_ms"], 4),
                         "roofline": roofline_block(kern3, st3["last_kernel_ms"], st3["sphere_tests"], W * H, len(sph3), "c3", args.mode, census=cen3, choice=ch3),
                         "moving_scene": moving3, "fast_mode": fast3,
                         "note": "frames, seeds and counters are the same bits through either form (DESIGN.md section 5; "
                                 "tests/test_gpu_parity.py test_baseline_configurations_at_full_size_bit_exact); bench.py --workload c3 is the full record"}
            except api.RtError as e:
                large = {"error": str(e)}
        # the in-library multi-device context with a communicator of one: what the frame-end gather path costs
        # when it has nothing to move (the N-GPU figure comes from the N > 1 runs)
        try:
            with api.RtContext(W, H, devices=[local_rank], tile_rows=TILE_ROWS) as m1:
                m1.set_scene(spheres); m1.set_camera(cam); m1.set_mode(mode)
                for _ in range(2):
                    m1.reset_async(); m1.render_pass(SPP, copy=False)
                t0 = time.perf_counter()
                for _ in range(8):                   # queued like the headline's frames (one stream: frame k + 1 starts when frame k is complete)
                    m1.reset_async(); m1.render_async(SPP)
                m1.throttle(0)
                dt = time.perf_counter() - t0
                in_library = {"path": "rt_create_multi(ngpus=1): one shard rendering straight into the frame (no communicator, no gather, no de-interleave); 8 frames queued with "
                                      "rt_reset_async + rt_render_async and ONE rt_throttle(0) at the end (since round 4; BENCH_r03's figure under this key timed blocking rt_render_pass calls)",
                              "method": "queued rt_render_async x 8 + rt_throttle(0)",
                              "ms_per_frame": round(dt / 8 * 1e3, 4), "frame_equals_headline": bool(np.array_equal(m1.read_pixels(), last_pixels))}
        except api.RtError as e:
            in_library = {"error": str(e)}

    my_local_rows = ctx.local_rows
    loop.close()
    # ---- N > 1: BASELINE configs[3] beside the headline -- C4 (Demo, 3840x2160, 256 spp), the configuration BASELINE built for the
    # 1/2/4/8 curve: a 1080p x 64 frame is short against its slowest wavefront, this one is not.  One frame at a time, every
    # gathered frame compared with the unsharded one on rank 0's GPU, exactly as the headline's ----
    c4_block = None
    if world > 1 and not args.no_extras and args.workload == "c2":
        W4, H4, SPP4 = 3840, 2160, 256
        sph4 = host.demo_scene()
        cam4 = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W4, H4)
        loop4 = FrameLoop(sph4, cam4, W4, H4, SPP4, 1)
        k4 = max(3, min(args.steps, 6))
        el4, kms4, _ = loop4.timed_region(1, k4, 1, with_events=True)
        st4 = loop4.frame_counters()
        samples4, shadow4 = (int(v) for v in all_ranks([st4["samples"], st4["shadow_rays"]], dist.ReduceOp.SUM))
        el4_max, kms4_max = all_ranks([el4, kms4], dist.ReduceOp.MAX)
        kms4_min = -all_ranks([-kms4], dist.ReduceOp.MAX)[0]
        if rank == 0:
            bad4 = int(loop4.mismatch.item())
            rays4 = samples4 + shadow4
            c4_block = {"workload": f"C4: Demo scene (6 spheres), {W4}x{H4}, {SPP4} spp, default seed stream (BASELINE configs[3])",
                        "steps": k4, "warmup": 1, "ms_per_step": round(el4_max / k4 * 1e3, 4), "value": round(rays4 * k4 / el4_max / 1e6, 1), "unit": "Mray/s",
                        "kernel": loop4.ctxs[0].last_kernel, "kernel_ms_max_rank": round(kms4_max, 4), "kernel_ms_min_rank": round(kms4_min, 4),
                        "shard_imbalance_max_over_min": round(kms4_max / max(kms4_min, 1e-9), 4),
                        "unsharded_kernel_ms_on_rank0_gpu": round(loop4.whole_kernel_ms, 4),
                        "speedup_vs_unsharded_kernel": round(loop4.whole_kernel_ms / (el4_max / k4 * 1e3), 3),
                        "every_gathered_frame_equals_unsharded": bad4 == 0, "gathered_frames_checked": loop4.frames_checked, "wrong_pixels": bad4,
                        "regime": "one frame at a time: gathered and assembled on rank 0 before the next starts", "rays_per_frame": rays4}
        loop4.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return 0

    if world > 1 and not args.no_extras and args.workload == "c2":
        # the in-library path on the same GPUs, as a child process (the ranks have released their communicator);
        # never allowed to take the headline down with it
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            time.sleep(2.0)
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--inproc-child", str(world), "--steps", str(max(5, args.steps // 2)),
                                  "--warmup", "2", "--mode", args.mode], env=env, capture_output=True, text=True, timeout=120)
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            in_library = json.loads(lines[-1]) if res.returncode == 0 and lines else {"error": (res.stderr or res.stdout)[-400:]}
        except (subprocess.TimeoutExpired, ValueError, OSError) as e:
            in_library = {"error": repr(e)[:400]}

    bound = None
    if not args.no_extras:
        try:
            bound = scaling_bound(api, mode, spheres, cam, W, H, SPP, kernel_ms if world == 1 else None, device=local_rank)
        except Exception as e:       # noqa: BLE001 -- a diagnostics figure never takes the headline down
            bound = {"error": repr(e)[:300]}
    # every shard of a 2 / 4 / 8-way split rendered ALONE on this GPU (tools/shard_prediction.py): what each GPU of an N-GPU run
    # would spend on the frame, imbalance and under-filled GPU included -- for this line's workload and for C4
    shards = None
    if not args.no_extras:              # (at N > 1 too: rank 0's GPU, alone by now -- the prediction stands beside the measurement it predicts)
        try:
            from tools import shard_prediction
            shards = {args.workload: shard_prediction.predict(api, mode, spheres, cam, W, H, SPP, whole_ms=kernel_ms if world == 1 else None,
                                                              whole_pixels=last_pixels, bound=bound, device=local_rank)}
            if args.workload == "c2":
                cam4 = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 3840, 2160)
                shards["c4"] = shard_prediction.predict(api, mode, host.demo_scene(), cam4, 3840, 2160, 256, frames=2, device=local_rank)
            shards["what"] = ("per N: each shard of the interleaved 8-row-tile split rendered alone on this GPU, steady state (kernel ms from the context's events); "
                              "predicted_ms_per_frame = slowest shard + measured de-interleave kernel + all packed pixels over one 153 GB/s xGMI link")
        except Exception as e:       # noqa: BLE001 -- a diagnostics figure never takes the headline down
            shards = {"error": repr(e)[:300]}
    rays = samples + shadow                      # primary + shadow, the metric's ray count
    ms_per_step = el1_max / args.steps * 1e3
    value = rays * args.steps / el1_max / 1e6
    # roofline of the dominant (only) kernel, per launch, from this rank's launches in the headline region
    # (HIP events on the stream the kernel is launched on; launches do not overlap there)
    census = None
    if world == 1 and "_pairs" in kernel_name:
        try:
            census = walk_census(api, spheres, cam, W, H, SPP)
        except api.RtError:            # (the census instance stages the whole tables in LDS: scenes beyond that have no executed-work count)
            census = None
    roofline = roofline_block(kernel_name, kernel_ms, st["sphere_tests"], my_local_rows * W, len(spheres), args.workload, args.mode,
                              census=census, choice=choice)
    if world > 1:
        roofline["traffic"] = None          # the committed counters describe the unsharded launch
        roofline.pop("executed", None)
    roofline["kernel_ms_max_rank"] = round(kernel_ms_max, 4)
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
                   "mode": args.mode, "regime": "one frame at a time (ms_per_step = ms/frame)",
                   "collective": None if world == 1 else f"gather to rank 0 ({backend}) + de-interleave kernel",
                   "every_gathered_frame_equals_unsharded": None if frames_ok is None else frames_ok["wrong_pixels"] == 0,
                   "gathered_frames_checked": None if frames_ok is None else frames_ok["frames_checked"],
                   "sharding": f"interleaved {TILE_ROWS}-row tiles x {world}",
                   "perfect_balance_estimate": bound,
                   "predicted_from_shards": shards,
                   "rays_per_frame": rays, "all_rays_per_frame": closest + shadow,
                   "Mray_s_all_rays": round((closest + shadow) * args.steps / el1_max / 1e6, 1),
                   "Msample_s": round(samples * args.steps / el1_max / 1e6, 1)},
        "roofline": roofline,
    }
    if elF_max is not None:
        line["frames_in_flight"] = {"frames": F, "ms_per_step": round(elF_max / args.steps * 1e3, 4),
                                    "value": round(rays * args.steps / elF_max / 1e6, 1), "unit": "Mray/s",
                                    "note": "the same K frames, F in flight on separate streams: throughput, not ms/frame"}
    if unseen is not None:
        line["unseen_passes"] = unseen
    if first is not None:
        line["first_frame"] = first
    if target is not None:
        line["north_star_target"] = target
    if large is not None:
        line["large_scene"] = large
    if other is not None:
        line["other_mode"] = other
    if c4_block is not None:
        line["c4"] = c4_block
    if in_library is not None:
        line["in_library_multi_gpu"] = in_library
    if world == 1 and not args.no_extras:
        try:
            line["host_inclusive"] = host_inclusive(spheres, cam, W, H, SPP, kernel_ms)
        except Exception as e:       # noqa: BLE001 -- a side figure never takes the headline down
            line["host_inclusive"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_cpu:
        base, cpu_out = cpu_baseline(spheres, cam, W, H, SPP)
        line["cpu_baseline"] = base
        if args.mode == "parity":
            line["config"]["matches_cpu_oracle_bit_exact"] = bool(np.array_equal(last_pixels, cpu_out["pixels"]))
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())
    return 0


if __name__ == "__main__":
    sys.exit(main())
