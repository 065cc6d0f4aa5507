#!/usr/bin/env python3
"""The reference's OWN inputs as a measured workload set: the Demo scene and the nine SimpleRT/Scene/*.scn
scenes at the reference's native window, 800 x 600 (SetupGL.cpp:32-33, Main.cpp:75-90).

    python tools/reference_scenes.py [--scenes demo,cornell,...] [--out profiles/r06_reference_scenes.jsonl] [--no-cpu]

Sphere arrays (as the reference's loader hands them to the kernel -- readScene's doubling included, Utility.cpp:120,154) and
camera origin / target come from the committed fixtures tests/golden/*.npz (made from the reference by tests/golden/make_golden.py):
/root/reference never travels to the GPU box.  Per scene, one JSON object:

  fused        64 passes in ONE launch per frame (the library's regime): the instance the library picks, median kernel ms of the
               steady frame, Gray/s (primary + shadow rays), roofline fraction (20 FLOP x reference sphere tests over the FP32 vector
               peak: executed work for the sweeps, the reference-equivalent rate for the hierarchy)
  per_pass     the reference's regime (Config.cpp:73-81, OpenCLConfig.cpp:407-515): one pass per call WITH the read-back of the
               frame, rt_render_pass(ctx, out, 1) x 64 -- wall ms per pass, passes per second
  alternative  forced A/B through the diagnostics library of the form the library did NOT pick (cooperative any-hit on / off,
               4-wavefront workgroups, sweep instead of hierarchy and the hierarchy's other table placements): a wrong pick shows
  cpu          the reference's kernel compiled as host C++ (oracle/_ref; the oracle where that build is absent) on the host
               cores, a bounded sample (fewer passes on the heavy scenes), and whether the GPU frame of the same passes equals it
  complex.scn additionally: pair steps and leaf visits per ray with and without the 783 zero-radius phantoms the loader's doubling puts
  at the origin (they are spheres to the reference and stay counted; the question is what they cost the walk)."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

W, H, SPP = 800, 600, 64
FIXTURES = {           # scene -> the fixture that carries its sphere array and camera
    "demo": "c1_demo_256x256_1spp", "simple": "simple_96x96_4spp", "cornell": "cornell_96x96_4spp", "cornell_large": "cornell_large_64x64_4spp",
    "caustic": "caustic_96x64_8spp", "caustic3": "caustic3_64x64_8spp", "demo_scn": "demo_scn_64x64_4spp", "complex": "complex_64x48_1spp",
    "cornell_test": "cornell_test_64x64_2spp", "complex_test": "complex_test_48x48_1spp",
}
SOURCE = {"demo": "DemoSpheres (Scene.cpp:5-12)", "demo_scn": "Scene/demo.scn"}
FP32_PEAK = 157.3e12


def load_scene(name):
    """(spheres, orig, target) of a reference scene from its committed fixture; also accepts a fixture's own name."""
    from raytracing_simple_amd import api
    z = np.load(os.path.join(GOLDEN, FIXTURES.get(name, name) + ".npz"))
    sph = np.ascontiguousarray(z["spheres"]).view(api.SPHERE_DT).copy()
    cam = np.asarray(z["camera"], np.float32)
    return sph, tuple(float(v) for v in cam[0:3]), tuple(float(v) for v in cam[3:6])


def steady(ctx, spp, frames=5, warm=3):
    ms = []
    for k in range(warm + frames):
        ctx.reset()
        ctx.render_pass(spp, copy=False)
        if k >= warm:
            ms.append(ctx.stats()["last_kernel_ms"])
    return statistics.median(ms)


def census(api, sph, cam, spp):
    """Pair steps / leaf visits per ray of the hierarchy walk on this scene (census instance, tables in LDS)."""
    with api.RtContext(W, H, diag=True) as c:
        c._check(c._lib.rt_debug_set_walk(c._h, 0, 0, 1))
        c.set_scene(sph)
        c.set_camera(cam)
        c.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
        c.render_pass(spp, copy=False)
        raw = (C.c_ulonglong * 32)()
        c._check(c._lib.rt_debug_counters_raw(c._h, raw))
        st = c.stats()
        cnt = np.zeros(4, np.uint32)
        c._check(c._lib.rt_debug_read_bvh(c._h, None, 0, cnt.ctypes.data_as(C.c_void_p)))
    rays = st["closest_rays"] + st["shadow_rays"]
    return {"pair_steps_per_ray": round(raw[21] / rays, 3), "leaf_visits_per_ray": round(raw[23] / rays, 3),
            "lanes_per_pair_step": round(raw[21] / max(raw[20], 1), 1), "always_list": int(cnt[0]), "leaves": int(cnt[1]), "stack_depth": int(cnt[2]), "rays": int(rays)}


def alternatives(api, sph, cam, picked_kernel, base_ms):
    """Forced A/B of the forms the library did not pick (diagnostics library, same frame, pixels compared with the pick's)."""
    lib = api.load_library(diag=True)
    out = []
    hier = "_pairs" in picked_kernel
    arms = []
    if hier:
        arms = [("plain sweep forced (no hierarchy)", dict(bvh=(0, 0))), ("whole hierarchy in LDS (rt_trace_parity_pairs)", dict(bvh=(56, 152 * 1024), walk=1)),
                ("hierarchy read from HBM / L2 (rt_trace_parity_pairs_g)", dict(bvh=(56, 1024), walk=1))]
    else:
        coop = "_coop" in picked_kernel
        arms = [("the library's pick, forced the same way (cooperative any-hit %s)" % ("on" if coop else "off"), dict(coop_min=(1 if coop else 0))),
                ("cooperative any-hit %s" % ("off" if coop else "on"), dict(coop_min=(0 if coop else 1))),
                ("4-wavefront workgroups" if "_w1" in picked_kernel else "1-wavefront workgroups", dict(wg=(4 if "_w1" in picked_kernel else 1), coop_min=(1 if coop else 0)))]
        if len(sph) >= 56:
            arms.append(("hierarchy forced", dict(walk=1)))
    want = None
    with api.RtContext(W, H, diag=True) as c0:
        c0.set_scene(sph); c0.set_camera(cam)
        want = c0.render_pass(SPP)
    for label, knobs in arms:
        try:
            with api.RtContext(W, H, diag=True) as c:
                if "bvh" in knobs:
                    c._check(lib.rt_debug_set_bvh(c._h, *knobs["bvh"]))
                if "walk" in knobs:
                    c._check(lib.rt_debug_set_walk(c._h, 0, 0, 1))
                if "coop_min" in knobs:
                    c._check(lib.rt_debug_set_coop_min(c._h, knobs["coop_min"]))
                if "wg" in knobs:
                    c._check(lib.rt_debug_set_wg_waves(c._h, knobs["wg"]))
                c.set_scene(sph); c.set_camera(cam)
                c.reset()
                px = c.render_pass(SPP)
                ms = steady(c, SPP, frames=5, warm=2)
                if label.startswith("the library's pick"):
                    base_ms = ms            # (the alternatives are compared with the pick measured the same way, in the same process)
                out.append({"form": label, "kernel": c.last_kernel, "kernel_ms": round(ms, 4), "vs_pick": round(ms / base_ms, 3),
                            "same_frame": bool(np.array_equal(px, want))})
        except api.RtError as e:
            out.append({"form": label, "error": str(e)[:160]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default=",".join(FIXTURES))
    ap.add_argument("--out", default="")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ab", action="store_true")
    args = ap.parse_args()
    from raytracing_simple_amd import api, host
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    import bench
    cores = bench.host_cores()
    for name in args.scenes.split(","):
        sph, orig, target = load_scene(name)
        cam = host.compute_camera(orig, target, W, H)
        rec = {"scene": name, "source": SOURCE.get(name, "Scene/%s.scn" % name), "spheres_seen_by_kernel": int(len(sph)),
               "zero_radius_phantoms": int((sph["rad"] == 0).sum()), "w": W, "h": H, "spp": SPP, "build_id": api.build_id()}
        with api.RtContext(W, H) as c:
            c.set_scene(sph); c.set_camera(cam)
            c.reset()
            px64 = c.render_pass(SPP)
            ms = steady(c, SPP)
            st = c.stats()
            kern, choice = c.last_kernel, c.scene_choice()
            rays = st["samples"] + st["shadow_rays"]
            flops = 20.0 * st["sphere_tests"]
            rec["fused"] = {"kernel": kern, "kernel_ms": round(ms, 4), "Gray_s": round(rays / ms / 1e6, 2), "rays_per_sample": round((st["closest_rays"] + st["shadow_rays"]) / st["samples"], 2),
                            "tests_per_sample": round(st["sphere_tests"] / st["samples"], 1),
                            "roofline_frac" if "_pairs" not in kern else "reference_equivalent_frac_not_executed": round(flops / (ms * 1e-3) / FP32_PEAK, 4),
                            "choice": choice["picked"]}
            # the reference's regime: one pass per call with the frame read back (page-locked, as the adapter does)
            out = np.zeros(W * H, np.uint32)
            c.pin_output(out)
            c.reset()
            per = []
            for k in range(SPP):
                t0 = time.perf_counter()
                c.render_pass(1, out=out)
                per.append(time.perf_counter() - t0)
            c.pin_output(None)
            med = statistics.median(per[8:])
            rec["per_pass"] = {"wall_ms_per_pass": round(med * 1e3, 4), "passes_per_s": round(1.0 / med, 1), "Msample_s": round(W * H / med / 1e6, 1),
                               "kernel": c.last_kernel, "frame_after_64_passes_equals_fused": bool(np.array_equal(out, px64)),
                               "fused_over_per_pass_time": round(ms / (med * 1e3 * SPP), 4)}
        if not args.no_cpu:
            heavy = len(sph) > 100
            cpu_spp = 2 if heavy else (16 if len(sph) > 12 else SPP)
            use_ref = O.ref_available()
            t0 = time.time()
            ref = O.ref_render_mt(sph, cam, W, H, cpu_spp, cores) if use_ref else O.render(sph, cam, W, H, cpu_spp, threads=cores)
            dt = time.time() - t0
            with api.RtContext(W, H) as c:
                c.set_scene(sph); c.set_camera(cam)
                got = c.render_pass(cpu_spp)
                stc = c.stats()
            rays_c = stc["samples"] + stc["shadow_rays"]
            rec["cpu"] = {"kind": "reference" if use_ref else "port", "cores": cores, "sample": f"{W}x{H} x {cpu_spp} passes", "seconds": round(dt, 2),
                          "Mray_s": round(rays_c / dt / 1e6, 2), "ms_per_pass": round(dt / cpu_spp * 1e3, 1),
                          "gpu_frame_of_the_same_passes_bit_exact": bool(np.array_equal(got, ref["pixels"]))}
            rec["gpu_over_cpu_per_pass"] = round((dt / cpu_spp * 1e3) / (ms / SPP), 0)
        if not args.no_ab:
            rec["alternative"] = alternatives(api, sph, cam, kern, ms)
        if "_pairs" in kern:
            try:
                rec["walk"] = {"with_phantoms": census(api, sph, cam, 8)}
                real = sph[sph["rad"] != 0]
                if len(real) != len(sph):
                    rec["walk"]["without_phantoms"] = census(api, real, cam, 8)
                    with api.RtContext(W, H) as c:
                        c.set_scene(real); c.set_camera(cam)
                        c.reset(); c.render_pass(SPP, copy=False)
                        ms_real = steady(c, SPP)
                        rec["walk"]["without_phantoms"].update({"kernel": c.last_kernel, "kernel_ms": round(ms_real, 4), "vs_with": round(ms_real / ms, 3),
                                                                "note": "a different INPUT (783 records fewer, other scene indices): diagnosis only, never rendered for a user"})
            except api.RtError as e:
                rec["walk"] = {"error": str(e)[:200]}
        line = json.dumps(rec)
        print(line, flush=True)
        if args.out:
            with open(os.path.join(ROOT, args.out), "a") as f:
                f.write(line + "\n")


if __name__ == "__main__":
    main()
