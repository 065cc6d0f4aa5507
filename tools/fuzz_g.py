#!/usr/bin/env python3
"""The plain sweep over a table beyond LDS (rt_trace_parity_g: the table through the scalar cache, four records per load) on fuzzed scenes:
`python tools/fuzz_g.py FIRST COUNT`.  A scene of tools/fuzz_parity.py's families 1 / 2 (ties, duplicates, NaN and infinite records, lights
anywhere) is padded to 9 729 ... 11 500 records with fillers -- records that are not numbers, infinite, zero or negative radii, tiny spheres far
away, small real spheres around the scene -- and shuffled, so that the scene's own spheres sit at any position of a group of four; the hierarchy
is switched off (rt_debug_set_bvh(0, 0)), 40x24 pixels, 3 passes in two launches.  Pixels, colour plane, seeds and the counters against the
oracle; prints the seeds that differ (none expected)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from raytracing_simple_amd import api, host  # noqa: E402

src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "api": api}
exec(src[src.index("def _fuzz_scene"):src.index('@pytest.mark.parametrize("seed"')], ns)


def padded(seed):
    sph, orig, target = ns["_fuzz_scene"](seed)
    rng = np.random.default_rng(500000 + seed)
    n = int(rng.integers(9729, 11501))
    fill = np.zeros(n - len(sph), api.SPHERE_DT)
    m = len(fill)
    fill["rad"] = rng.uniform(0.05, 0.8, m).astype(np.float32)
    fill["p"] = rng.uniform(-120, 120, (m, 3)).astype(np.float32)
    fill["c"] = rng.uniform(0.1, 0.9, (m, 3)).astype(np.float32)
    fill["refl"] = rng.choice([api.DIFF, api.SPEC, api.REFR], m)
    kind = rng.integers(0, 10, m)
    with np.errstate(all="ignore"):
        fill["p"][kind == 0, int(rng.integers(0, 3))] = np.float32("nan")
        fill["rad"][kind == 1] = np.float32("inf")
        fill["p"][kind == 2, int(rng.integers(0, 3))] = np.float32("inf")
        fill["rad"][kind == 3] = 0.0
        fill["rad"][kind == 4] *= np.float32(-1.0)
        fill["p"][kind == 5] *= np.float32(1e4)
        fill["rad"][kind == 6] = np.float32("nan")
    allsph = np.concatenate([sph, fill])
    rng.shuffle(allsph)
    return allsph, orig, target


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    w, h = 40, 24
    bad = []
    kernels = {}
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        for seed in range(first, first + count):
            sph, orig, target = padded(seed)
            cam = host.compute_camera(orig, target, w, h)
            with np.errstate(all="ignore"):
                want = O.render(sph, cam, w, h, 3, threads=16)
                ctx.set_scene(sph)
                ctx.set_camera(cam)
                ctx.reset()
                ctx.render_pass(1)
                px = ctx.render_pass(2)
            g, o = ctx.stats(), want["stats"]
            kernels[ctx.last_kernel] = kernels.get(ctx.last_kernel, 0) + 1
            same = (np.array_equal(px, want["pixels"]) and np.array_equal(ctx.read_colors().view(np.uint32), want["colors"].view(np.uint32))
                    and np.array_equal(ctx.read_seeds(), want["seeds"])
                    and (g["samples"], g["closest_rays"], g["shadow_rays"], g["sphere_tests"], g["rng_draws"]) ==
                    (o["samples"], o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"]))
            if not same:
                bad.append(seed)
                print("MISMATCH seed", seed, flush=True)
            if (seed - first) % 50 == 49:
                print("...", seed - first + 1, "scenes", flush=True)
    print("fuzz_g: build", api.build_id(diag=True), "seeds", first, "..", first + count - 1, "kernels", kernels, "mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
