#!/usr/bin/env python3
"""Many fuzzed scenes (the generator of tests/test_gpu_parity.py) through the HIP path and the oracle:
python tools/fuzz_parity.py FIRST COUNT -- prints the seeds that differ (none expected)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
from raytracing_simple_amd import api, host
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "api": api}
exec(src[src.index("def _fuzz_scene"):src.index('@pytest.mark.parametrize("seed"')], ns)
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    sph, orig, target = ns["_fuzz_scene"](seed)
    w, h, spp = [(40, 24, 3), (33, 17, 2), (64, 32, 5), (25, 40, 4), (96, 64, 2), (17, 9, 9)][seed % 6]
    cam = host.compute_camera(orig, target, w, h)
    with np.errstate(all="ignore"):
        want = O.render(sph, cam, w, h, spp)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        px = ctx.render_pass(spp); col = ctx.read_colors(); sd = ctx.read_seeds(); st = ctx.stats()
    o = want["stats"]
    same = (np.array_equal(px, want["pixels"]) and np.array_equal(col.view(np.uint32), want["colors"].view(np.uint32))
            and np.array_equal(sd, want["seeds"]) and
            (st["closest_rays"], st["shadow_rays"], st["sphere_tests"], st["rng_draws"]) ==
            (o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"]))
    if not same:
        bad.append(seed)
        print("MISMATCH seed", seed, "n", len(sph), (w, h, spp), "pixel diffs", int((px != want["pixels"]).sum()),
              "nan in oracle colours", bool(np.isnan(want["colors"]).any()), flush=True)
print("fuzz seeds", first, "..", first + count - 1, "mismatches:", bad)
