#!/usr/bin/env python3
"""A performance fuzz of the library's choice of form: scenes of varied counts, radius distributions (one class, two classes, log-uniform over
three decades), layouts (a slab over a ground sphere, clusters, a closed box) and material mixes, each rendered (640x360, 2 passes) by the
library's own pick, with the hierarchy forced and with the sweep forced; a line per scene, `worse_than_best` = pick's time over the best form's.
`python tools/choice_fuzz.py FIRST COUNT [directory of another build of the libraries]` (diagnostics library);
RT_CHOICE_FUZZ_FAMILY=small: scenes of 4 ... 250 spheres with 1 ... 12 lights, pick against the hierarchy forced and the plain / cooperative sweep
with one / four wavefronts per workgroup; RT_CHOICE_FUZZ_FAMILY=lights: 50 ... 3000 spheres of which 10 ... 300 are lights.  Frames are compared bit for bit on the way."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host  # noqa: E402


def scene(seed):
    rng = np.random.default_rng(900000 + seed)
    n = int(10 ** rng.uniform(2.0, 4.3))
    layout = ["slab", "clusters", "box"][seed % 3]
    radii = ["one", "two", "log3", "two_rev"][(seed // 3) % 4]
    sph = np.zeros(n + 8, api.SPHERE_DT)
    if radii == "one":
        r = rng.uniform(0.5, 1.5, n)
    elif radii == "two":
        r = np.where(rng.random(n) < 0.7, rng.uniform(0.02, 0.06, n), rng.uniform(1.0, 2.5, n))
    elif radii == "two_rev":
        r = np.where(rng.random(n) < 0.25, rng.uniform(0.02, 0.06, n), rng.uniform(1.0, 2.5, n))
    else:
        r = 10 ** rng.uniform(-2, 1, n)
    if layout == "slab":
        p = np.stack([rng.uniform(-70, 70, n), r + rng.uniform(0, 8, n), rng.uniform(-70, 70, n)], 1)
    elif layout == "clusters":
        k = int(rng.integers(3, 12))
        cen = np.stack([rng.uniform(-60, 60, k), rng.uniform(3, 25, k), rng.uniform(-60, 60, k)], 1)
        p = cen[rng.integers(0, k, n)] + rng.normal(0, 4.0, (n, 3))
        p[:, 1] = np.abs(p[:, 1]) + r
    else:
        p = np.stack([rng.uniform(5, 95, n), rng.uniform(2, 75, n), rng.uniform(10, 160, n)], 1)
    sph["rad"][:n] = r.astype(np.float32)
    sph["p"][:n] = p.astype(np.float32)
    sph["c"][:n] = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    mix = [[api.DIFF], [api.DIFF, api.DIFF, api.SPEC, api.REFR], [api.SPEC, api.REFR]][(seed // 12) % 3]
    sph["refl"][:n] = rng.choice(mix, n)
    m = n
    if layout == "box":
        walls = [(1e4, (1e4 + 1, 40.8, 81.6)), (1e4, (-1e4 + 99, 40.8, 81.6)), (1e4, (50, 40.8, 1e4)), (1e4, (50, 40.8, -1e4 + 270)), (1e4, (50, 1e4, 81.6)),
                 (1e4, (50, -1e4 + 81.6, 81.6))]
        for rad, pos in walls:
            sph["rad"][m], sph["p"][m], sph["c"][m] = rad, pos, (.75, .75, .75)
            m += 1
        sph["rad"][m], sph["p"][m], sph["e"][m] = 7, (50, 66.6, 81.6), (12, 12, 12)
        m += 1
        orig, target = (50.0, 45.0, 205.6), (50.0, 44.957388, 204.6)
    else:
        sph["rad"][m], sph["p"][m], sph["c"][m] = 1000.0, (0, -1000, 0), (.75, .75, .75)
        m += 1
        sph["rad"][m], sph["p"][m], sph["e"][m] = 9.0, (0, 70, 0), (14, 14, 14)
        m += 1
        orig, target = host.DEMO_ORIG, host.DEMO_TARGET
    return sph[:m].copy(), orig, target, {"n": int(m), "layout": layout, "radii": radii, "materials": len(mix)}


def small_scene(seed):
    """4 ... 250 spheres over a ground sphere, 1 ... 12 of them lights: the band where plain / cooperative sweep, one / four wavefronts per
    workgroup and (from 56 spheres) the hierarchy compete."""
    rng = np.random.default_rng(910000 + seed)
    n = int(10 ** rng.uniform(0.6, 2.4))
    sph = np.zeros(n + 1, api.SPHERE_DT)
    r = rng.uniform(1.0, 6.0, n)
    sph["rad"][:n] = r.astype(np.float32)
    sph["p"][:n] = np.stack([rng.uniform(-45, 45, n), r + rng.uniform(0, 20, n), rng.uniform(-45, 45, n)], 1).astype(np.float32)
    sph["c"][:n] = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    mix = [[api.DIFF], [api.DIFF, api.DIFF, api.SPEC, api.REFR], [api.SPEC, api.REFR]][seed % 3]
    sph["refl"][:n] = rng.choice(mix, n)
    lights = rng.choice(n, min(n, int(rng.integers(1, 13))), replace=False)
    sph["e"][lights] = rng.uniform(2, 12, (len(lights), 3)).astype(np.float32)
    sph["refl"][lights] = api.DIFF
    sph["rad"][n], sph["p"][n], sph["c"][n] = 1000.0, (0, -1000, 0), (.75, .75, .75)
    return sph, host.DEMO_ORIG, host.DEMO_TARGET, {"n": int(n + 1), "lights": int(len(lights)), "materials": len(mix), "family": "small"}


def lights_scene(seed):
    """50 ... 3000 spheres of one size class over a ground sphere, 10 ... 300 of them lights: every diffuse hit asks every light."""
    rng = np.random.default_rng(920000 + seed)
    n = int(10 ** rng.uniform(1.7, 3.5))
    sph = np.zeros(n + 1, api.SPHERE_DT)
    r = rng.uniform(0.5, 2.0, n)
    sph["rad"][:n] = r.astype(np.float32)
    sph["p"][:n] = np.stack([rng.uniform(-60, 60, n), r + rng.uniform(0, 15, n), rng.uniform(-60, 60, n)], 1).astype(np.float32)
    sph["c"][:n] = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    sph["refl"][:n] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n)
    lights = rng.choice(n, min(n // 2, int(10 ** rng.uniform(1.0, 2.5))), replace=False)
    sph["e"][lights] = rng.uniform(1, 6, (len(lights), 3)).astype(np.float32)
    sph["refl"][lights] = api.DIFF
    sph["rad"][n], sph["p"][n], sph["c"][n] = 1000.0, (0, -1000, 0), (.75, .75, .75)
    return sph, host.DEMO_ORIG, host.DEMO_TARGET, {"n": int(n + 1), "lights": int(len(lights)), "family": "lights"}


def run(sph, cam, w, h, spp, how):
    with api.RtContext(w, h, diag=True) as ctx:
        if how == "sweep":
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        elif how == "hierarchy":
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 31 * 1024))
            ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        elif how in ("plain", "coop", "plain_w4", "coop_w4"):
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
            ctx._check(ctx._lib.rt_debug_set_coop_min(ctx._h, 1 if how.startswith("coop") else 0))
            ctx._check(ctx._lib.rt_debug_set_wg_waves(ctx._h, 4 if how.endswith("_w4") else 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ts, px = [], None
        for k in range(8):                       # (a scene's first launches may measure its forms against each other: 8 frames, the median of the last 3)
            ctx.reset()
            px = ctx.render_pass(16 if k < 5 else spp)
            ts.append(ctx.stats()["last_kernel_ms"])
        return statistics.median(ts[5:]), ctx.last_kernel, px


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    if len(sys.argv) > 3:            # another build of the two libraries (a directory): the A/B against the library before a change
        alt = os.path.abspath(sys.argv[3])
        api.lib_path = lambda diag=False: os.path.join(alt, "librt_hip_diag.so" if diag else "librt_hip.so")
    w, h, spp = 640, 360, 2
    small = os.environ.get("RT_CHOICE_FUZZ_FAMILY") == "small"
    many_lights = os.environ.get("RT_CHOICE_FUZZ_FAMILY") == "lights"
    for seed in range(first, first + count):
        sph, orig, target, what = small_scene(seed) if small else (lights_scene(seed) if many_lights else scene(seed))
        cam = host.compute_camera(orig, target, w, h)
        res = {}
        px0 = None
        for how in (("pick", "hierarchy", "plain", "coop", "plain_w4", "coop_w4") if small else ("pick", "hierarchy", "sweep")):
            try:
                ms, k, px = run(sph, cam, w, h, spp, how)
            except api.RtError as e:
                res[how] = {"error": str(e)[:80]}
                continue
            res[how] = {"ms": round(ms, 3), "kernel": k}
            if px0 is None:
                px0 = px
            elif not np.array_equal(px0, px):
                res[how]["FRAME_DIFFERS"] = True
        best = min(v["ms"] for v in res.values() if "ms" in v)
        what.update({"seed": seed, "forms": res, "worse_than_best": round(res["pick"]["ms"] / best, 2)})
        print(json.dumps(what), flush=True)


if __name__ == "__main__":
    main()
