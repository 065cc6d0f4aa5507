#!/usr/bin/env python3
"""Many fuzzed scenes (the generator of tests/test_gpu_parity.py) through the HIP path and the oracle:
python tools/fuzz_parity.py FIRST COUNT [FAMILY] -- prints the seeds that differ (none expected).
RT_FUZZ_BVH=1|2 forces the hierarchy of large scenes on every scene (1 = walk per call, 2 = walk as lane state;
diagnostics library); family 4 = hundreds of spheres in clusters, radii over three orders of magnitude; family 6 = two size
classes / radii over three decades at hundreds to thousands of spheres (the cut between tree and always-list)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
from raytracing_simple_amd import api, host
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "api": api}
exec(src[src.index("def _fuzz_scene"):src.index('@pytest.mark.parametrize("seed"')], ns)
def family2(seed):
    """Ties and containment: exact duplicates of spheres (equal distances: the lowest index must win),
    concentric shells, spheres that touch, up to 8 lights, the camera inside nested glass."""
    rng = np.random.default_rng(100000 + seed)
    base = int(rng.integers(1, 12))
    sph = np.zeros(base, api.SPHERE_DT)
    sph["rad"] = rng.uniform(1.0, 30.0, base).astype(np.float32)
    sph["p"] = rng.uniform(-50, 50, (base, 3)).astype(np.float32)
    sph["c"] = rng.uniform(0.1, 0.95, (base, 3)).astype(np.float32)
    sph["refl"] = rng.choice([api.DIFF, api.SPEC, api.REFR], base)
    parts = [sph]
    dup = sph[rng.integers(0, base, int(rng.integers(1, 6)))].copy()          # exact duplicates, other materials
    dup["refl"] = rng.choice([api.DIFF, api.SPEC, api.REFR], len(dup))
    dup["c"] = rng.uniform(0.1, 0.95, (len(dup), 3)).astype(np.float32)
    parts.append(dup)
    shell = sph[rng.integers(0, base, int(rng.integers(1, 5)))].copy()        # concentric shells
    shell["rad"] = (shell["rad"] * rng.choice([0.5, 0.999, 1.0000001, 1.5, 2.0], len(shell))).astype(np.float32)
    shell["refl"] = api.REFR
    parts.append(shell)
    touch = sph[:1].copy()                                                    # a sphere touching sphere 0
    touch["p"][0] = sph["p"][0] + np.float32([sph["rad"][0] + 5.0, 0, 0])
    touch["rad"] = 5.0
    parts.append(touch)
    allsph = np.concatenate(parts)
    rng.shuffle(allsph)
    for j in rng.choice(len(allsph), min(int(rng.integers(0, 9)), len(allsph)), replace=False):
        allsph["e"][j] = rng.uniform(1.0, 30.0, 3).astype(np.float32)
    inside = allsph["p"][0] + np.float32(0.1) * allsph["rad"][0]
    orig = inside if seed % 2 else rng.uniform(-90, 90, 3).astype(np.float32)
    target = rng.uniform(-20, 20, 3).astype(np.float32)
    return allsph, tuple(float(v) for v in orig), tuple(float(v) for v in target)


def family3(seed):
    """Family 1 scaled by 10^-3 ... 10^4 (radii, centres, camera): the fixed EPSILON = 0.01 then
    cuts into the geometry (tiny scenes) or drowns in rounding error (huge ones), and the camera
    may sit a hair above a surface."""
    sph, orig, target = ns["_fuzz_scene"](seed)
    rng = np.random.default_rng(200000 + seed)
    k = np.float32(10.0 ** rng.uniform(-3, 4))
    sph = sph.copy()
    sph["rad"] = sph["rad"] * k
    sph["p"] = sph["p"] * k
    orig = tuple(float(np.float32(v) * k) for v in orig)
    target = tuple(float(np.float32(v) * k) for v in target)
    if seed % 3 == 0 and len(sph):
        j = int(rng.integers(0, len(sph)))
        up = np.float32([0, 1, 0]) * (sph["rad"][j] * np.float32(1.0 + 10.0 ** rng.uniform(-7, -2)))
        orig = tuple(float(v) for v in (sph["p"][j] + up))
    return sph, orig, target


def family4(seed):
    """Scenes the hierarchy is built for, made nasty: 60..600 spheres in a few clusters, radii log-uniform over three
    orders of magnitude (the largest stay outside the tree), exact duplicates, a cluster far away, the whole scene
    scaled by 10^-2 ... 10^3, the camera inside a cluster or inside a glass sphere."""
    rng = np.random.default_rng(300000 + seed)
    n = int(rng.choice([60, 90, 150, 260, 400, 600]))
    k = int(rng.integers(1, 6))
    centres = rng.uniform(-60, 60, (k, 3))
    spread = rng.uniform(3, 40, k)
    which = rng.integers(0, k, n)
    sph = np.zeros(n, api.SPHERE_DT)
    sph["p"] = (centres[which] + rng.normal(0, 1, (n, 3)) * spread[which, None]).astype(np.float32)
    sph["rad"] = (10.0 ** rng.uniform(-1.0, 2.0, n) * 0.5).astype(np.float32)
    sph["c"] = rng.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    sph["refl"] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n)
    dup = rng.integers(0, n // 2, 6)
    sph[n - 6:] = sph[dup]
    sph["refl"][n - 6:] = rng.choice([api.DIFF, api.SPEC, api.REFR], 6)
    if seed % 3 == 0:
        sph["p"][n // 2:n // 2 + 5] += np.float32(5000.0)
    for j in rng.choice(n, int(rng.integers(1, 4)), replace=False):
        sph["e"][j] = rng.uniform(2.0, 25.0, 3).astype(np.float32)
    scale = np.float32(10.0 ** rng.uniform(-2, 3)) if seed % 2 else np.float32(1.0)
    sph["p"] *= scale
    sph["rad"] *= scale
    j = int(rng.integers(0, n))
    orig = sph["p"][j] + np.float32(0.3) * sph["rad"][j] if seed % 4 == 1 else (centres[0] + rng.normal(0, 1, 3) * 90).astype(np.float32) * scale
    target = (centres[int(rng.integers(0, k))] * scale).astype(np.float32)
    return sph, tuple(float(v) for v in orig), tuple(float(v) for v in target)


def family5(seed):
    """What round 6's builders special-case, made nasty: the reference loader's pattern -- a block of zero-radius records at one point in FRONT
    of the real ones (Utility.cpp:120,154), the point being the origin or the very spot the camera looks at --, records that repeat earlier ones
    in centre and radius^2 with other materials / negated radius (rt_bvh.hip mark_duplicates), and spheres far SMALLER than the rest (radii down
    to 10^-5 of the median, some exactly zero, scattered where rays pass: their boxes are grown at build time, bvh_half_width)."""
    rng = np.random.default_rng(400000 + seed)
    n_real = int(rng.choice([70, 120, 200, 330]))
    n_ph = int(rng.choice([0, 8, 40, n_real]))
    real = np.zeros(n_real, api.SPHERE_DT)
    real["p"] = rng.uniform(-40, 40, (n_real, 3)).astype(np.float32)
    real["rad"] = rng.uniform(0.5, 4.0, n_real).astype(np.float32)
    real["c"] = rng.uniform(0.05, 0.95, (n_real, 3)).astype(np.float32)
    real["refl"] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n_real)
    real["rad"][0], real["p"][0], real["refl"][0] = 1000.0, (0, -1040, 0), api.DIFF
    small = rng.choice(np.arange(2, n_real), n_real // 5, replace=False)
    real["rad"][small] = (real["rad"][small] * 10.0 ** rng.uniform(-5, -1, len(small))).astype(np.float32)
    real["rad"][small[: len(small) // 4]] = 0.0
    for j in rng.choice(n_real, int(rng.integers(1, 4)), replace=False):
        real["e"][j] = rng.uniform(2.0, 25.0, 3).astype(np.float32)
    k = max(2, n_real // 8)
    src = rng.integers(1, n_real // 2, k)
    dst = n_real - 1 - np.arange(k)
    real[dst] = real[src]
    real["refl"][dst] = rng.choice([api.DIFF, api.SPEC, api.REFR], k)
    real["c"][dst] = rng.uniform(0.05, 0.95, (k, 3)).astype(np.float32)
    real["rad"][dst[::3]] = -real["rad"][dst[::3]]
    spot = np.zeros(3, np.float32) if seed % 2 else real["p"][int(rng.integers(1, n_real // 2))].copy()
    ph = np.zeros(n_ph, api.SPHERE_DT)
    ph["p"] = spot
    sph = np.concatenate([ph, real])
    orig = rng.uniform(-70, 70, 3).astype(np.float32)
    target = spot if seed % 3 else rng.uniform(-10, 10, 3).astype(np.float32)
    return sph, tuple(float(v) for v in orig), tuple(float(v) for v in target)


def family6(seed):
    """What round 6's CUT special-cases (rt_bvh.hip build_bvh_tables: the always-list holds only spheres of the scene's own size): scenes of two
    size classes -- 200 ... 4000 small spheres among 9 ... 800 spheres 20 ... 100 times their size, either class the majority --, or radii over
    three decades; a ground sphere and walls far larger than both; some of the large ones exact repeats, not finite, or far away (the extent is
    a quantile: they must not stretch it); the camera anywhere, also inside a large sphere."""
    rng = np.random.default_rng(600000 + seed)
    n_small = int(10 ** rng.uniform(2.3, 3.6))
    n_large = int(rng.choice([9, 12, 40, 150, 400, 800]))
    r0 = float(10 ** rng.uniform(-2, 0))
    ratio = float(rng.uniform(20, 100))
    n = n_small + n_large + 3
    sph = np.zeros(n, api.SPHERE_DT)
    if seed % 5 == 4:
        sph["rad"][:n - 3] = (r0 * 10.0 ** rng.uniform(0, 3, n - 3)).astype(np.float32)
    else:
        sph["rad"][:n_small] = (r0 * rng.uniform(0.6, 1.6, n_small)).astype(np.float32)
        sph["rad"][n_small:n - 3] = (r0 * ratio * rng.uniform(0.6, 1.6, n_large)).astype(np.float32)
    span = r0 * ratio * float(rng.uniform(8, 40))
    sph["p"][:n - 3] = np.stack([rng.uniform(-span, span, n - 3), rng.uniform(0, span / 4, n - 3), rng.uniform(-span, span, n - 3)], 1).astype(np.float32)
    sph["c"] = rng.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    sph["refl"] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n)
    sph["rad"][n - 3], sph["p"][n - 3], sph["refl"][n - 3] = 1000.0 * span, (0, -1000.0 * span, 0), api.DIFF          # ground
    sph["rad"][n - 2], sph["p"][n - 2], sph["refl"][n - 2] = 500.0 * span, (-501.0 * span - span, 0, 0), api.DIFF      # a wall
    sph["rad"][n - 1], sph["p"][n - 1], sph["e"][n - 1] = span / 6, (0, span, 0), (9, 9, 9)                            # the light
    big = np.arange(n_small, n - 3)
    if len(big) >= 12:
        sph[big[:3]] = sph[big[3:6]]                                            # exact repeats among the large ones, other materials
        sph["refl"][big[:3]] = rng.choice([api.DIFF, api.SPEC, api.REFR], 3)
        with np.errstate(all="ignore"):
            sph["p"][big[6], 0] = np.float32("nan")
            sph["rad"][big[7]] = np.float32("inf")
        sph["p"][big[8:11]] += np.float32(1e4 * span)                           # far away
    for j in rng.choice(n - 3, int(rng.integers(0, 3)), replace=False):
        sph["e"][j] = rng.uniform(2.0, 25.0, 3).astype(np.float32)
    perm = rng.permutation(n)
    sph = sph[perm]
    if seed % 4 == 1 and n_large:
        j = int(np.argmax(np.where(np.isfinite(sph["rad"]) & (sph["rad"] < span), sph["rad"], 0)))
        orig = sph["p"][j] + np.float32(0.3) * sph["rad"][j]
    else:
        orig = np.float32([rng.uniform(-span, span), rng.uniform(span / 8, span / 2), rng.uniform(-span, span)])
    target = np.float32([rng.uniform(-span / 4, span / 4), 0, rng.uniform(-span / 4, span / 4)])
    return sph, tuple(float(v) for v in orig), tuple(float(v) for v in target)


first, count = int(sys.argv[1]), int(sys.argv[2])
bvh_form = int(os.environ.get("RT_FUZZ_BVH", "0"))
gen = {"2": family2, "3": family3, "4": family4, "5": family5, "6": family6}.get(sys.argv[3] if len(sys.argv) > 3 else "1", ns["_fuzz_scene"])
bad = []
kernels = {}
for seed in range(first, first + count):
    sph, orig, target = gen(seed)
    w, h, spp = [(40, 24, 3), (33, 17, 2), (64, 32, 5), (25, 40, 4), (96, 64, 2), (17, 9, 9)][seed % 6]
    spp *= int(os.environ.get("RT_FUZZ_SPP_SCALE", "1"))          # (8: frames of 16 .. 72 passes -- long enough for a first frame to price its tiles, rt_api.hip launch_priced)
    cam = host.compute_camera(orig, target, w, h)
    with np.errstate(all="ignore"):
        want = O.render(sph, cam, w, h, spp)
    with api.RtContext(w, h, diag=bvh_form != 0) as ctx:
        if bvh_form:
            if os.environ.get("RT_FUZZ_TREE_SHAPE"):                     # 0: the device build's fixed shape (default: by surface area, on the host)
                ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, int(os.environ["RT_FUZZ_TREE_SHAPE"])))
            # RT_FUZZ_LDS = a small LDS budget in bytes: trees whose whole tables exceed it but whose pairs fit run on the instance that
            # stages the pairs and reads the slots from L2 (..._pairs_m), larger ones on ..._pairs_g -- `kernels` below says which ran
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, int(os.environ.get("RT_FUZZ_LDS", 152 * 1024))))
            ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))      # RT_FUZZ_BVH != 0: the hierarchy forced
        ctx.set_scene(sph); ctx.set_camera(cam)
        px = ctx.render_pass(spp); col = ctx.read_colors(); sd = ctx.read_seeds(); st = ctx.stats()
        kernels[ctx.last_kernel] = kernels.get(ctx.last_kernel, 0) + 1
    o = want["stats"]
    same = (np.array_equal(px, want["pixels"]) and np.array_equal(col.view(np.uint32), want["colors"].view(np.uint32))
            and np.array_equal(sd, want["seeds"]) and
            (st["closest_rays"], st["shadow_rays"], st["sphere_tests"], st["rng_draws"]) ==
            (o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"]))
    if not same:
        bad.append(seed)
        print("MISMATCH seed", seed, "n", len(sph), (w, h, spp), "pixel diffs", int((px != want["pixels"]).sum()),
              "nan in oracle colours", bool(np.isnan(want["colors"]).any()), flush=True)
print("fuzz seeds", first, "..", first + count - 1, "mismatches:", bad, "kernels:", kernels)
