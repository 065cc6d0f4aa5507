#!/usr/bin/env python3
"""Convergence of the progressive estimate (frame writer + convergence tooling).

    python tools/convergence.py [--scene demo|c16|FILE.scn] [--w 480 --h 270] [--max-spp 4096]
                                [--ref-spp 65536] [--out gpurun_out/convergence]

Renders one scene progressively through the C ABI (rt_render_pass continues the running average,
as repeated Config::updateRendering() calls do in the reference), writes a PPM at every power of
two and prints one JSON line per checkpoint: PSNR of the packed pixels and RMSE of the float colour
plane against a long reference render, for both arithmetic modes, plus the PSNR of fast against
parity at equal spp (the north star's fast-mode gate)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host, scenes  # noqa: E402


def load(name):
    if name == "demo":
        return host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET
    if name == "c16":
        return scenes.demo_plus(16)
    return host.read_scene(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="demo")
    ap.add_argument("--w", type=int, default=480)
    ap.add_argument("--h", type=int, default=270)
    ap.add_argument("--max-spp", type=int, default=4096)
    ap.add_argument("--ref-spp", type=int, default=65536)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    sph, orig, target = load(args.scene)
    cam = host.compute_camera(orig, target, args.w, args.h)

    def progressive(mode, checkpoints):
        shots = {}
        with api.RtContext(args.w, args.h) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            ctx.set_mode(mode)
            done = 0
            for spp in checkpoints:
                px = ctx.render_pass(spp - done)          # continues the running average
                done = spp
                shots[spp] = (px.copy(), ctx.read_colors().copy())
        return shots

    cps = [1 << k for k in range(0, args.max_spp.bit_length()) if (1 << k) <= args.max_spp]
    ref = progressive(api.RT_MODE_PARITY, [args.ref_spp])[args.ref_spp]
    host.write_ppm(os.path.join(args.out, f"{args.scene}_ref_{args.ref_spp}spp.ppm"), ref[0], args.w, args.h)
    par = progressive(api.RT_MODE_PARITY, cps)
    fast = progressive(api.RT_MODE_FAST, cps)
    for spp in cps:
        host.write_ppm(os.path.join(args.out, f"{args.scene}_parity_{spp}spp.ppm"), par[spp][0], args.w, args.h)
        rm = lambda a: float(np.sqrt(np.mean((a.astype(np.float64) - ref[1].astype(np.float64)) ** 2)))
        print(json.dumps({"scene": args.scene, "w": args.w, "h": args.h, "spp": spp, "ref_spp": args.ref_spp,
                          "psnr_parity_vs_ref_db": round(host.psnr(par[spp][0], ref[0]), 2),
                          "psnr_fast_vs_ref_db": round(host.psnr(fast[spp][0], ref[0]), 2),
                          "psnr_fast_vs_parity_same_spp_db": round(host.psnr(fast[spp][0], par[spp][0]), 2),
                          "rmse_colour_parity": round(rm(par[spp][1]), 6),
                          "rmse_colour_fast": round(rm(fast[spp][1]), 6)}), flush=True)


if __name__ == "__main__":
    main()
