#!/usr/bin/env python3
"""In-process A/B of kernel instances (guide rule 24: interleaved rounds in ONE process).

    python tools/ab_bench.py [--configs c2,c3,c5,c16] [--modes 0,1,rt_trace_parity_persist,...] [--rounds 5]

Each mode renders the same frame (0 / 1 = what the library picks in parity / fast mode; a kernel symbol = that
instance of the diagnostics library); parity-arithmetic instances must produce the same pixels as mode 0, fast ones
are reported with their PSNR against mode 0."""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402

CONFIGS = {
    "c2": (lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 1920, 1080, 64),
    "c16": (lambda: scenes.demo_plus(16), 1920, 1080, 64),
    "c9": (lambda: scenes.demo_plus(9), 1920, 1080, 64),
    "c10": (lambda: scenes.demo_plus(10), 1920, 1080, 64),
    "c11": (lambda: scenes.demo_plus(11), 1920, 1080, 64),
    "c12": (lambda: scenes.demo_plus(12), 1920, 1080, 64),
    "c32": (lambda: scenes.demo_plus(32), 1920, 1080, 64),
    "c3": (lambda: scenes.random_spheres(1024), 1920, 1080, 16),
    "c5": (lambda: scenes.mirror_box(64), 1920, 1080, 64),
    "c64": (lambda: scenes.random_spheres(64), 1920, 1080, 64),
    "c256": (lambda: scenes.random_spheres(256), 1920, 1080, 32),
    "c4": (lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), 3840, 2160, 256),
    # scenes for the two shipped instances no BASELINE configuration reaches: the 4-wavefront cooperative sweep, the walk over tables in HBM / L2
    "box120": (lambda: scenes.mirror_box(120), 1920, 1080, 8),
    "box300": (lambda: scenes.mirror_box(300), 1920, 1080, 8),
    "box700": (lambda: scenes.mirror_box(700), 1920, 1080, 4),
    "r2048": (lambda: scenes.random_spheres(2048), 1920, 1080, 8),
    "r3000": (lambda: scenes.random_spheres(3000), 1920, 1080, 8),
    "r8192": (lambda: scenes.random_spheres(8192), 1920, 1080, 4),
    "r65536": (lambda: scenes.random_spheres(65536), 1920, 1080, 4),
    "r262144": (lambda: scenes.random_spheres(262144), 1920, 1080, 4),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--coop-min", type=int, default=-1, help="override the cooperative any-hit threshold (0 = off)")
    ap.add_argument("--persist", default="", help="comma list of 0/1: sweep persistent-wavefront instances")
    ap.add_argument("--mat-lds", type=int, default=-1, help="override the LDS byte limit for staging materials")
    ap.add_argument("--skip-pixels", action="store_true", help="launches leave the packed pixels alone (rt_set_pixel_write 0)")
    ap.add_argument("--gates", default="", help="comma list of regeneration gates to sweep (mode list then = base modes)")
    ap.add_argument("--orders", default="", help="comma list of 0/1: natural tile order / heavy tiles first")
    ap.add_argument("--unseen", action="store_true", help="time launches on passes not rendered before (three launches continuing the image after each "
                                                            "reset) instead of the same frame again: what a schedule derived from the last launch's costs "
                                                            "is worth when those costs are predictions, not answers")
    args = ap.parse_args()
    modes = [int(m) if m.lstrip("-").isdigit() else api.instance_mode(m) for m in args.modes.split(",")]
    gates = [int(g) for g in args.gates.split(",")] if args.gates else [None]
    persists = [int(g) for g in args.persist.split(",")] if args.persist else [None]
    orders = [int(g) for g in args.orders.split(",")] if args.orders else [None]
    variants = [(m, g, q, o) for m in modes for g in gates for q in persists for o in orders]
    for cname in args.configs.split(","):
        maker, w, h, spp = CONFIGS[cname]
        if args.spp:
            spp = args.spp
        sph, orig, target = maker()
        cam = host.compute_camera(orig, target, w, h)
        with api.RtContext(w, h, diag=True) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            times = {v: [] for v in variants}
            pix, stats, kernels = {}, {}, {}
            lib = api.load_library(diag=True)
            if args.mat_lds >= 0:
                lib.rt_debug_set_mat_lds_limit(ctx._h, args.mat_lds)
            if args.coop_min >= 0:
                lib.rt_debug_set_coop_min(ctx._h, args.coop_min)
            if args.skip_pixels:
                ctx.set_pixel_write(False)
            for r in range(args.rounds + 1):
                for v in variants:
                    m, g, q, o = v
                    ctx.set_mode(m)
                    lib.rt_debug_set_regen_gate(ctx._h, 0 if g is None else g)
                    if q is not None:
                        lib.rt_debug_set_persist(ctx._h, q)
                    if o is not None:
                        lib.rt_debug_set_tile_order(ctx._h, o)
                        if o:                               # (the order is dropped when the knob moves: one untimed launch sorts again)
                            ctx.reset()
                            ctx.render_pass(spp)
                    ctx.reset()
                    px = ctx.render_pass(spp)
                    st = ctx.stats()
                    kernels[v] = ctx.last_kernel
                    if r == 0:
                        pix[v], stats[v] = px, st            # warm-up round: keep outputs only
                    elif args.unseen:
                        for _ in range(3):
                            ctx.render_pass(spp, copy=False)
                            times[v].append(ctx.stats()["last_kernel_ms"])
                    else:
                        times[v].append(st["last_kernel_ms"])
            base = pix[variants[0]]
            for v in variants:
                m, g, q, o = v
                st = stats[v]
                rays = st["samples"] + st["shadow_rays"]
                med, mn = statistics.median(times[v]), min(times[v])
                same = bool(np.array_equal(pix[v], base))
                print(json.dumps({"config": cname, "mode": m, "kernel": kernels[v], "gate": g, "persist": q, "heavy_first": o, "unseen_passes": bool(args.unseen), "ms_median": round(med, 4), "ms_min": round(mn, 4),
                                  "Gray_s": round(rays / med / 1e6, 2), "same_as_first": same,
                                  "psnr_vs_first": None if same else round(host.psnr(pix[v], base), 2),
                                  "tests_per_sample": round(st["sphere_tests"] / st["samples"], 2),
                                  "alg_TFLOPs": round(20 * st["sphere_tests"] / med / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    main()
