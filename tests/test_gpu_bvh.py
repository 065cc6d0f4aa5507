"""Large scenes: the hierarchy over the small spheres (csrc/rt_device.h BvhTables, rt_walk.inc.h).  It only selects
which spheres a ray is tested against, so everything must stay BIT-EXACT: frames, colour plane, seeds and the work
counters against the oracle, with the hierarchy forced on small and adversarial scenes too."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import _oracle as O
from raytracing_simple_amd import api, host, scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bvh_check  # noqa: E402

pytestmark = pytest.mark.gpu


def _render(sph, cam, w, h, spp, bvh_min=1, form=1, mode=api.RT_MODE_PARITY, passes=None, by_area=1, estimate=1, inst=None):
    if inst:
        mode = api.instance_mode(inst)           # a kernel instance of the diagnostics library by its symbol
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_choice_estimate(ctx._h, estimate))
        ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, by_area))
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, bvh_min, 152 * 1024))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, form))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.set_mode(mode)
        px = None
        for n in (passes or [spp]):
            px = ctx.render_pass(n)
        return {"pixels": px, "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats(),
                "pick": ctx._lib.rt_debug_bvh_pick(ctx._h)}


L2_WALKS = ["rt_trace_parity_pairs_g", "rt_trace_parity_pairs_gq", "rt_trace_parity_pairs_gt", "rt_trace_parity_pairs_gp"]


def _same(got, want):
    assert np.array_equal(got["pixels"], want["pixels"])
    assert np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))
    assert np.array_equal(got["seeds"], want["seeds"])
    g, o = got["stats"], want["stats"]
    assert (g["samples"], g["closest_rays"], g["shadow_rays"], g["sphere_tests"], g["rng_draws"]) == \
           (o["samples"], o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"])


@pytest.mark.parametrize("maker", [lambda: scenes.random_spheres(1024), lambda: scenes.random_spheres(97),
                                   lambda: scenes.mirror_box(64), lambda: scenes.demo_plus(16),
                                   lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET)])
def test_device_built_tables_are_a_valid_hierarchy(maker):
    """All three builds: the device's fixed shape (leaf ranges halved), the host's of a full scene upload (the shape chosen by surface
    area, leaves of up to 8; the default below 1500 tree spheres) and the device's by surface area (cuts between whole leaves; what
    updates and large uploads get) -- every leaf reached once from the root pair the header names, every sphere in one leaf, inside
    every box above it, lowest scene indices right, the stack deep enough."""
    sph, _, _ = maker()
    sph = api.as_spheres(sph)
    leaves, area = {}, {}
    for by_area in (0, 1, 2):
        with api.RtContext(64, 64, diag=True) as ctx:
            ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, by_area))
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
            ctx.set_scene(sph)
            b = bvh_check.read_bvh(ctx)
            pk = bvh_check.read_packed(ctx)
        assert b is not None
        assert bvh_check.check_structure(sph, b) == []
        if b["n_leaves"] > 1:               # the diagnostics library promotes the top of EVERY tree and packs its pairs (rt_bvh.hip promote_top, pack_pairs)
            assert b["root"] == 0
            assert pk is not None and bvh_check.check_packed(b, pk) == []
        assert b["n_always"] + sum(1 for i in b["index"][b["n_always"]:] if i != 0xffffffff) == len(bvh_check.first_of_equals(sph))
        leaves[by_area] = b["n_leaves"]
        area[by_area] = bvh_check.sum_of_box_areas(b)
    assert leaves[0] <= leaves[1] <= 2 * leaves[0]              # partial leaves only where they pay (below 128 tree spheres both are the device's)
    assert leaves[2] == leaves[0]                               # the device cuts between whole leaves
    n_tree = len(sph) - b["n_always"]
    if n_tree >= 128:                                           # (below that every build is the halved shape)
        assert area[2] < area[0] and area[2] < 1.1 * area[1]    # what the walk pays for: the boxes a ray can meet


@pytest.mark.parametrize("maker,w,h,spp", [
    (lambda: scenes.random_spheres(1024), 96, 64, 3),
    (lambda: scenes.random_spheres(300), 80, 60, 4),
    (lambda: scenes.mirror_box(64), 64, 64, 4),
    (lambda: scenes.mirror_box(200), 48, 48, 3),
    (lambda: scenes.demo_plus(16), 96, 64, 4),
])
def test_the_walk_equals_the_oracle(maker, w, h, spp):
    """the hierarchy forced on scenes of every kind (open, boxed, too small to get one by default): frames, colour plane,
    seeds and counters are the oracle's"""
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    got = _render(sph, cam, w, h, spp)
    _same(got, want)
    _same(_render(sph, cam, w, h, spp, by_area=0), want)        # the device build's fixed shape


def test_walk_and_plain_sweep_agree_ray_by_ray():
    """Rays of the scene itself (camera rays, bounce rays off the spheres, rays towards the lights) through the walk and
    through the plain sweep, one lane per ray: closest hit -- distance bits and sphere index; shadow rays -- first
    blocking index."""
    for maker, (w, h) in ((lambda: scenes.random_spheres(1024), (128, 72)), (lambda: scenes.mirror_box(96), (96, 64))):
        sph, orig, target = maker()
        r = bvh_check.agreement(sph, host.compute_camera(orig, target, w, h), w, h)
        assert r is not None and r["closest_hits"] > 10000 and r["shadow_blocked"] > 1000, r
        assert r["closest_differ"] == 0 and r["shadow_differ"] == 0, r


def _adversarial(seed):
    """Duplicated spheres (the reference's loader doubles them: exact ties), zero and negative radii, concentric and
    heavily overlapping spheres, a camera inside a glass sphere, non-finite records, a far-away cluster."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([40, 70, 130]))
    sph = np.zeros(n, api.SPHERE_DT)
    sph["rad"] = rng.uniform(0.5, 6.0, n).astype(np.float32)
    sph["p"] = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
    sph["c"] = rng.uniform(0.1, 0.95, (n, 3)).astype(np.float32)
    sph["refl"] = rng.integers(0, 3, n)
    sph["rad"][0], sph["p"][0], sph["refl"][0] = 1000.0, (0, -1030, 0), 0          # ground
    sph["rad"][1], sph["p"][1], sph["e"][1] = 8.0, (0, 45, 0), (10, 10, 10)         # light
    half = n // 2
    dup = rng.integers(2, half, 8)
    sph[half:half + 8] = sph[dup]                                                   # exact duplicates, higher index
    sph["rad"][half + 8] = 0.0
    sph["rad"][half + 9] = -3.0                                                      # rad*rad is what the test uses
    sph["p"][half + 10] = sph["p"][half + 11]                                        # concentric
    sph["rad"][half + 12] = np.float32("nan")
    sph["p"][half + 13, 1] = np.float32("inf")
    sph["p"][half + 14] = (4000.0, 10.0, -3000.0)                                    # far-away member of the tree
    orig = (float(sph["p"][3, 0]), float(sph["p"][3, 1]), float(sph["p"][3, 2]) + 0.5) if seed % 2 else (10.0, 30.0, 70.0)
    if seed % 2:
        sph["refl"][3], sph["rad"][3] = 2, 5.0                                       # the camera sits inside glass
    return sph, orig, (0.0, 5.0, 0.0)


@pytest.mark.parametrize("seed", range(6))
def test_adversarial_scenes_equal_the_oracle_with_the_hierarchy_forced(seed):
    sph, orig, target = _adversarial(seed)
    w, h, spp = 72, 48, 3
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    _same(_render(sph, cam, w, h, spp), want)
    _same(_render(sph, cam, w, h, spp, by_area=0), want)
    for inst in L2_WALKS:                                                           # the walks that read their tables from HBM / L2, every form of them
        _same(_render(sph, cam, w, h, spp, inst=inst), want)
    r = bvh_check.agreement(sph, cam, w, h, 60000)
    assert r["closest_differ"] == 0 and r["shadow_differ"] == 0, r


def _with_repeats(seed):
    """A scene in which later records repeat earlier ones bit for bit in centre and radius^2 but NOT in material (a repeated diffuse sphere that
    is glass, black, a light), a repeated light, a repeated ground (always-list), a record with the negated radius (same radius^2) and the
    reference loader's own pattern: a block of zero-radius records at the origin in FRONT of everything (Utility.cpp:120,154)."""
    rng = np.random.default_rng(100 + seed)
    n_real, n_ph = 90, 40
    real = np.zeros(n_real, api.SPHERE_DT)
    real["rad"] = rng.uniform(0.8, 5.0, n_real).astype(np.float32)
    real["p"] = rng.uniform(-30, 30, (n_real, 3)).astype(np.float32)
    real["p"][:, 1] = np.abs(real["p"][:, 1])
    real["c"] = rng.uniform(0.1, 0.95, (n_real, 3)).astype(np.float32)
    real["refl"] = rng.integers(0, 3, n_real)
    real["rad"][0], real["p"][0], real["refl"][0] = 1000.0, (0, -1000, 0), 0         # ground
    real["rad"][1], real["p"][1], real["e"][1], real["refl"][1] = 8.0, (0, 45, 0), (10, 10, 10), 0          # light
    for k, src in enumerate(rng.integers(2, 40, 12)):                                # repeats with OTHER materials, at higher indices
        dst = 60 + k
        real[dst] = real[src]
        real["refl"][dst] = (int(real["refl"][src]) + 1 + k % 2) % 3
        real["c"][dst] = (0.05, 0.9, 0.05)
        if k % 4 == 0:
            real["e"][dst] = (3, 3, 3)                                               # a repeat that is a light (it is sampled as one; never hit)
        if k % 3 == 0:
            real["rad"][dst] = -real["rad"][dst]                                     # same radius^2
    real[75] = real[1]                                                               # the light, repeated
    real[76] = real[0]                                                               # the ground, repeated (always-list)
    real["c"][76] = (0.9, 0.1, 0.1)
    phantoms = np.zeros(n_ph, api.SPHERE_DT)
    return np.concatenate([phantoms, real]), (20.0, 40.0, 90.0), (0.0, 8.0, 0.0)


@pytest.mark.parametrize("seed", range(3))
def test_repeated_records_stay_out_of_the_hierarchy_and_change_no_bit(seed):
    """rt_bvh.hip mark_duplicates: a record that repeats an earlier one in centre and radius^2 is never a ray's answer (the reference keeps the
    first of equals), so the hierarchy leaves it out -- whatever its material says.  Frames, colour plane, seeds and counters are the oracle's
    through every builder; the tables hold exactly the first of each set of equals; and a device-resident update that ends a repeat (the record
    moves away) or creates one brings the record back / takes it out."""
    sph, orig, target = _with_repeats(seed)
    w, h, spp = 80, 56, 4
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    for by_area in (1, 0, 2):
        _same(_render(sph, cam, w, h, spp, by_area=by_area), want)
    keep = bvh_check.first_of_equals(sph)
    assert len(keep) == len(sph) - 39 - 12 - 2                                     # 39 of the 40 phantoms, the 12 repeats, the repeated light and ground
    for by_area in (0, 1, 2):
        with api.RtContext(64, 64, diag=True) as ctx:
            ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, by_area))
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
            ctx.set_scene(sph)
            b = bvh_check.read_bvh(ctx)
        assert bvh_check.check_structure(sph, b) == []
        assert b["n_always"] == 1                                                    # ONE ground
    # updates: record 100 (a repeat) moves away -> it must be rendered; record 50 becomes a repeat of record 45 -> out again
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 152 * 1024))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        assert np.array_equal(ctx.render_pass(spp), want["pixels"])
        moved = api.as_spheres(sph).copy()
        moved["p"][100] = (5.0, 12.0, 20.0)
        moved["rad"][100] = 6.0
        ctx.update_spheres(100, moved[100:101])
        ctx.reset()
        assert np.array_equal(ctx.render_pass(spp), O.render(moved, cam, w, h, spp)["pixels"])
        moved[50] = moved[45]
        moved["refl"][50] = 1
        ctx.update_spheres(50, moved[50:51])
        ctx.reset()
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        _same(got, O.render(moved, cam, w, h, spp))
        assert bvh_check.check_structure(moved, bvh_check.read_bvh(ctx)) == []


def test_measured_choice_changes_no_bit_and_probes_split_a_blocking_call():
    """form 0 = the library's behaviour.  Where the surface-area estimate of the uploaded tree is clear it decides and nothing
    is measured (one launch); inside its band -- or with the estimate switched off -- the first launches of a new scene time
    both forms (hierarchy warm, hierarchy timed, sweep warm, sweep timed) and the faster renders the rest.  Whatever is
    picked, and however the passes are split, the frame is the oracle's."""
    sph, orig, target = scenes.random_spheres(200)
    w, h, spp = 96, 64, 20
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    clear = _render(sph, cam, w, h, spp, bvh_min=64, form=0)
    _same(clear, want)
    assert clear["pick"] == 1 and clear["stats"]["launches"] == 1       # 200 spheres scattered on a plane: predicted ratio 0.43
    one = _render(sph, cam, w, h, spp, bvh_min=64, form=0, estimate=0)
    _same(one, want)
    assert one["pick"] in (1, 2)                     # a blocking call of >= 16 passes waits for the verdict
    assert one["stats"]["launches"] == 5             # each form warm and timed, then the rest
    many = _render(sph, cam, w, h, spp, bvh_min=64, form=0, passes=[1, 1, 2, 3, 13], estimate=0)
    _same(many, want)
    box, borig, btarget = scenes.mirror_box(200)     # a packed box of 200: predicted 0.97, inside the band -> measured
    bcam = host.compute_camera(borig, btarget, w, h)
    banded = _render(box, bcam, w, h, spp, bvh_min=64, form=0)
    _same(banded, O.render(box, bcam, w, h, spp))
    assert banded["pick"] in (1, 2) and banded["stats"]["launches"] == 5
    c5, corig, ctarget = scenes.mirror_box(64)       # C5's box: predicted 1.65 -> the sweep, unmeasured
    ccam = host.compute_camera(corig, ctarget, w, h)
    swept = _render(c5, ccam, w, h, spp, bvh_min=56, form=0)
    _same(swept, O.render(c5, ccam, w, h, spp))
    assert swept["pick"] == 2 and swept["stats"]["launches"] == 1
    small = _render(*((lambda s: (s[0], host.compute_camera(s[1], s[2], w, h)))(scenes.demo_plus(16))), w, h, 4, bvh_min=64, form=0)
    assert small["pick"] == 0 and small["stats"]["launches"] == 1      # no hierarchy below bvh_min: nothing to measure
    big, orig, target = scenes.random_spheres(1700)                    # from 1500 spheres on the answer is known: no probe
    bcam = host.compute_camera(orig, target, 48, 32)
    with api.RtContext(48, 32, diag=True) as ctx:
        ctx.set_scene(big)
        ctx.set_camera(bcam)
        px = ctx.render_pass(20)
        assert ctx.stats()["launches"] == 1 and ctx.last_kernel.startswith("rt_trace_parity_pairs") and ctx.scene_choice()["picked"] is None
        assert np.array_equal(px, O.render(big, bcam, 48, 32, 20)["pixels"])


def test_moving_spheres_rebuild_the_hierarchy_on_the_stream():
    sph, orig, target = scenes.random_spheres(160)
    sph = api.as_spheres(sph).copy()
    w, h, spp = 80, 48, 3
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        for step in range(3):
            sph["p"][10:40, 0] += np.float32(1.5)
            sph["p"][10:40, 1] += np.float32(0.25 * step)
            ctx.update_spheres(10, sph[10:40])
            ctx.reset()
            got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
            _same(got, O.render(sph, cam, w, h, spp))
            b = bvh_check.read_bvh(ctx)
            assert bvh_check.check_structure(sph, b) == []


def test_an_update_of_the_whole_scene_stays_on_the_stream():
    """rt_update_spheres_async that rewrites EVERY record (the usual animated scene) is still an update: a device build on the
    stream -- full leaves, the shape chosen by surface area by the device's own kernel --, never the host build of rt_set_scene,
    which waits for the staging buffer's previous copy (the range must not decide)."""
    sph, orig, target = scenes.random_spheres(300)
    sph = api.as_spheres(sph).copy()
    w, h, spp = 64, 48, 2
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        shaped = bvh_check.read_bvh(ctx)
        sph["p"][:, 0] += np.float32(0.5)
        ctx.update_spheres(0, sph)                      # first = 0, count = n: the whole scene
        b = bvh_check.read_bvh(ctx)
        n_tree = b["n_slots"] - b["n_always"]
        assert b["n_leaves"] == (int((np.abs(sph["rad"]) <= 16 * np.median(np.abs(sph["rad"]))).sum()) + 7) // 8
        assert n_tree == 8 * b["n_leaves"]              # (the host's build of the upload makes partial leaves, the device's never)
        assert bvh_check.check_structure(sph, b) == []
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        _same(got, O.render(sph, cam, w, h, spp))


def test_updates_get_the_shape_by_surface_area():
    """A moving scene's frames walk a tree shaped by surface area as an upload's do (the device's own build): fewer pair steps per
    ray than through the halved shape, the same bits."""
    sph, orig, target = scenes.random_spheres(1024)
    sph = api.as_spheres(sph).copy()
    w, h, spp = 160, 96, 2
    cam = host.compute_camera(orig, target, w, h)
    moved = sph.copy()
    moved["p"][2:, 0] += np.float32(0.75)
    want = O.render(moved, cam, w, h, spp)
    steps = {}
    for by_area in (0, 1):
        with api.RtContext(w, h, diag=True) as ctx:
            ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, by_area))
            ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            ctx.update_spheres(0, moved)
            b = bvh_check.read_bvh(ctx)
            assert bvh_check.check_structure(moved, b) == []
            got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
            _same(got, want)
            ctx.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
            ctx.reset()
            ctx.render_pass(spp, copy=False)
            steps[by_area] = bvh_check.counters_raw(ctx)[21]        # pair steps, summed over lanes
    assert steps[1] < 0.95 * steps[0]


def test_fast_mode_with_the_hierarchy_is_as_close_as_fast_mode_without():
    """Fused arithmetic changes single bounces, and on a scene with mirrors and glass a changed bounce changes a path:
    fast mode's distance from parity mode is a property of the scene (the 50 dB gate is quoted on the Demo scene).
    With the hierarchy it must be what it is with the plain sweep."""
    sph, orig, target = scenes.random_spheres(400)
    w, h, spp = 160, 96, 16
    cam = host.compute_camera(orig, target, w, h)
    par = _render(sph, cam, w, h, spp)
    plain = _render(sph, cam, w, h, spp, bvh_min=0, mode=api.RT_MODE_FAST)
    base = host.psnr(plain["pixels"], par["pixels"])
    assert base >= 30.0
    fast = _render(sph, cam, w, h, spp, mode=api.RT_MODE_FAST)
    assert fast["stats"]["samples"] == par["stats"]["samples"]
    assert host.psnr(fast["pixels"], par["pixels"]) >= min(50.0, base - 3.0)


def test_last_kernel_names_the_instance_the_scene_got():
    """rt_last_kernel: which instance the library picked (what a profiler will list)."""
    w, h = 64, 48
    with api.RtContext(w, h) as ctx:
        assert ctx.last_kernel == ""
        ctx.set_scene(host.demo_scene())
        ctx.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h))
        # six spheres: cooperative any-hit or not is MEASURED on the host's own launches -- coop warm (one launch), coop timed (16 passes), plain warm,
        # plain timed; then the faster one
        seen = []
        for _ in range(19):
            ctx.render_pass(2)
            seen.append(ctx.last_kernel)
        assert seen[:9] == ["rt_trace_parity_coop_w1"] * 9 and seen[9:18] == ["rt_trace_parity_w1"] * 9, seen
        assert seen[18] in ("rt_trace_parity_w1", "rt_trace_parity_coop_w1")
        ctx.set_mode(api.RT_MODE_FAST)
        ctx.render_pass(2)
        assert ctx.last_kernel == seen[18].replace("parity", "fast")            # (the verdict is the scene's, whatever the arithmetic mode)
        ctx.set_mode(api.RT_MODE_PARITY)
        sph, orig, target = scenes.demo_plus(16)
        ctx.set_scene(sph)
        ctx.render_pass(2)
        assert ctx.last_kernel == "rt_trace_parity_coop_w1"
        sph, orig, target = scenes.random_spheres(600)     # 600 spheres scattered on a plane: the estimate is clear, nothing is measured
        ctx.set_scene(sph)
        ctx.set_camera(host.compute_camera(orig, target, w, h))
        ctx.reset()
        for _ in range(3):
            ctx.render_pass(1)
            assert ctx.last_kernel == "rt_trace_parity_pairs"
        ch = ctx.scene_choice()
        assert ch["picked"] == "hierarchy" and ch["hierarchy_ms_per_pass"] == 0 and ch["sweep_ms_per_pass"] == 0
        sph, orig, target = scenes.mirror_box(200)         # a packed box of 200: inside the estimate's band, so it is measured
        ctx.set_scene(sph)
        ctx.set_camera(host.compute_camera(orig, target, w, h))
        ctx.reset()
        ctx.render_pass(1)                    # the first two launches of such a scene walk the hierarchy (warm, timed),
        assert ctx.last_kernel == "rt_trace_parity_pairs"
        ctx.render_pass(1)
        assert ctx.last_kernel == "rt_trace_parity_pairs"
        ctx.render_pass(1)                    # the next two sweep; the faster form renders the rest
        assert ctx.last_kernel == "rt_trace_parity_coop"
        ctx.render_pass(1)
        assert ctx.last_kernel == "rt_trace_parity_coop"
        assert ctx.scene_choice()["picked"] in (None, "hierarchy", "sweep")   # (asked without blocking: may still be open)
        ctx.render_pass(40)
        assert ctx.last_kernel in ("rt_trace_parity_pairs", "rt_trace_parity_coop")
        ch = ctx.scene_choice()                                    # both probes finished before that blocking call returned
        assert ch["picked"] == ("hierarchy" if ctx.last_kernel.endswith("pairs") else "sweep")
        assert ch["hierarchy_ms_per_pass"] > 0 and ch["sweep_ms_per_pass"] > 0


def test_multi_device_context_with_a_large_scene():
    """Every shard builds its own hierarchy; the FIRST shard measures hierarchy against sweep and the others follow its
    verdict, so one kernel instance renders the whole frame; the assembled frames are the oracle's."""
    sph, orig, target = scenes.random_spheres(300)
    w, h = 96, 72
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, devices=[0, 0, 0]) as ctx:
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        done = 0
        for n in (1, 1, 2, 1, 20):               # the first four launches are the first shard's probes
            px = ctx.render_pass(n)
            done += n
            want = O.render(sph, cam, w, h, done)
            assert np.array_equal(px, want["pixels"])
        st = ctx.stats()
        assert (st["samples"], st["sphere_tests"]) == (want["stats"]["samples"], want["stats"]["sphere_tests"])
        assert np.array_equal(ctx.read_seeds(), want["seeds"])


def test_headline_call_on_a_large_scene():
    """rt_render(scene, cam, out, w, h, spp): a blocking call, so a new large scene is probed inside it; repeated
    calls with the same scene reuse the verdict.  Either way the frame is the oracle's."""
    sph, orig, target = scenes.random_spheres(220)
    w, h, spp = 88, 56, 18
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)["pixels"]
    for _ in range(3):
        assert np.array_equal(api.render(sph, cam, w, h, spp), want)
    box, orig, target = scenes.mirror_box(100)
    cam = host.compute_camera(orig, target, w, h)
    assert np.array_equal(api.render(box, cam, w, h, spp), O.render(box, cam, w, h, spp)["pixels"])
    api.load_library().rt_release_cache()


def _adversarial_rays(sph, rng, n):
    """Rays a path tracer produces once in 10^8 samples, by the thousand: exact zeros and denormals in the
    direction, origins on box planes and sphere surfaces, inside spheres, far away; tangent rays; directions that are
    not unit vectors, zero, or not finite."""
    c = np.ascontiguousarray(sph["p"]).astype(np.float64)
    r = np.abs(sph["rad"].astype(np.float64))
    ok = np.isfinite(c).all(1) & np.isfinite(r) & (r < 500)
    c, r = c[ok], r[ok]
    lo, hi = c.min(0) - 5, c.max(0) + 5
    rays = np.zeros((n, 8), np.float32)
    for i in range(n):
        kind = i % 12
        j = int(rng.integers(0, len(c)))
        o = rng.uniform(lo, hi)
        d = rng.normal(0, 1, 3)
        d /= np.linalg.norm(d)
        if kind == 1:                                   # axis-aligned: two exact zeros
            d = np.zeros(3); d[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])
        elif kind == 2:                                 # one exact zero
            d[int(rng.integers(0, 3))] = 0.0; d /= np.linalg.norm(d)
        elif kind == 3:                                 # denormal / tiny components
            k = int(rng.integers(0, 3)); d[k] = rng.choice([1e-40, -1e-42, 1e-30, -1e-20, 1e-12])
        elif kind == 4:                                 # origin exactly on a box plane of sphere j, sliding along it
            k = int(rng.integers(0, 3)); o = c[j].copy(); o[k] += rng.choice([-1.0, 1.0]) * r[j]
            o[(k + 1) % 3] -= 3 * r[j]
            d = np.zeros(3); d[(k + 1) % 3] = 1.0
        elif kind == 5:                                 # a bounce ray: from the surface of sphere j (or from inside it)
            nrm = rng.normal(0, 1, 3); nrm /= np.linalg.norm(nrm)
            o = c[j] + nrm * r[j] * rng.choice([1.0, 1.0, 0.5, 0.0])
        elif kind == 6:                                 # far away, aimed at the scene
            o = d * -rng.choice([1e4, 1e6, 1e9, 1e12, 1e19, 1e25]) + c[j]
        elif kind == 7:                                 # not a unit vector / not finite
            d = d * rng.choice([0.5, 2.0, 1e-3, 1e3, 0.0, 1.0005, 0.9995])
            if i % 5 == 0:
                d[int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
        elif kind == 8:                                 # tangent to sphere j, a hair inside or outside
            t = np.cross(d, rng.normal(0, 1, 3)); t /= np.linalg.norm(t)
            o = c[j] + t * r[j] * (1 + rng.choice([-1e-6, 1e-6, -1e-7, 1e-7, 0.0])) - d * rng.uniform(0, 40)
        elif kind == 9:                                 # aimed at a centre: a certain hit
            d = c[j] - o; d /= max(np.linalg.norm(d), 1e-30)
        elif kind == 10:                                # origin non-finite
            o[int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf])
        rays[i, 0:3] = o
        rays[i, 3] = rng.choice([1e20, 10.0, 50.0, 0.5, 1e-3])
        rays[i, 4:7] = d
        rays[i, 7:8].view(np.uint32)[0] = i & 1        # odd: shadow ray
    return rays


@pytest.mark.parametrize("maker", [lambda: scenes.random_spheres(1024), lambda: scenes.mirror_box(120), lambda: _adversarial(3)])
def test_walk_equals_sweep_for_adversarial_rays(maker):
    sph = api.as_spheres(maker()[0])
    rng = np.random.default_rng(len(sph))
    rays = _adversarial_rays(sph, rng, 60000)
    with api.RtContext(32, 32, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 152 * 1024))
        ctx.set_scene(sph)
        out = np.zeros((len(rays), 4), np.uint32)
        with np.errstate(all="ignore"):
            ctx._check(ctx._lib.rt_debug_walk_rays(ctx._h, rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p)))
    differ = np.nonzero((out[:, 0] != out[:, 2]) | (out[:, 1] != out[:, 3]))[0]
    assert len(differ) == 0, (differ[:5], rays[differ[:5]], out[differ[:5]])
    closest = out[0::2]
    assert (closest[:, 0] != 0xffffffff).sum() > 1000          # the harness is not vacuous: thousands of hits
    assert (out[1::2, 0] < len(sph)).sum() > 1000               # ... and of blocked shadow rays


def _many_spheres(n, seed=7):
    """n small spheres in a slab above a ground sphere, one light: beyond what LDS holds for n > ~9000."""
    rng = np.random.default_rng(seed)
    sph = np.zeros(n, api.SPHERE_DT)
    sph["rad"] = rng.uniform(0.3, 1.2, n).astype(np.float32)
    sph["p"] = np.stack([rng.uniform(-90, 90, n), rng.uniform(0.5, 9, n), rng.uniform(-90, 90, n)], 1).astype(np.float32)
    sph["c"] = rng.uniform(0.1, 0.9, (n, 3)).astype(np.float32)
    sph["refl"] = rng.choice([api.DIFF, api.DIFF, api.SPEC, api.REFR], n)
    sph["rad"][0], sph["p"][0], sph["refl"][0], sph["c"][0] = 1000.0, (0, -1000, 0), api.DIFF, (.75, .75, .75)
    sph["rad"][1], sph["p"][1], sph["e"][1], sph["refl"][1] = 9.0, (0, 70, 0), (14, 14, 14), api.DIFF
    return sph, host.DEMO_ORIG, host.DEMO_TARGET


@pytest.mark.parametrize("n", [2000, 5000, 9500, 30000])
def test_scenes_beyond_lds(n):
    """More spheres than the LDS budget holds (the hierarchy's whole tables stop at ~1100 spheres, the sweep's at ~2500: four workgroups per CU): while the PAIRS
    still fit (to ~3200 spheres) they are staged and only the slots read from HBM / L2 (rt_trace_parity_pairs_m, round 5); beyond, the
    walk reads pairs and slots from HBM / L2 (rt_trace_parity_pairs_g); beyond 8192 spheres in the tree the tables are built on the
    host.  Frames, seeds and counters are still the oracle's."""
    sph, orig, target = _many_spheres(n)
    w, h, spp = 48, 32, 2
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp, threads=16)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        assert ctx.last_kernel == ("rt_trace_parity_pairs_m" if n <= 3000 else "rt_trace_parity_pairs_g")
        _same(got, want)
        if n <= 3000:                   # ... and the same scene with everything read from HBM / L2 (the budget set below the pairs)
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 56, 1024))
            ctx.reset()
            got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
            assert ctx.last_kernel == "rt_trace_parity_pairs_g"
            _same(got, want)
        b = bvh_check.read_bvh(ctx)
        assert bvh_check.check_structure(api.as_spheres(sph), b) == []
        # the plain sweep over the table in HBM / L2 (what a scene without a hierarchy gets)
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        ctx.reset()
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        assert ctx.last_kernel == ("rt_trace_parity_coop" if 16 * n + 128 <= 40 * 1024 else "rt_trace_parity_g")      # (staged while 4 workgroups fit a CU)
        _same(got, want)


@pytest.mark.parametrize("n", [9729, 9730, 9731, 9732, 12001])
def test_plain_sweep_through_the_scalar_cache(n):
    """rt_trace_parity_g (round 6): the plain sweep over a table beyond LDS reads it four records per scalar load, the next four in flight
    while these are tested, then two and one for what is left -- every residue of the table size modulo 4, records that are not numbers or
    infinite among the spheres (they are tested like the others and never hit), three lights (shadow sweeps that leave early, at every
    position of a group), a pass continued by a second launch."""
    sph, orig, target = _many_spheres(n, seed=n)
    rng = np.random.default_rng(n)
    odd = rng.choice(np.arange(2, n), 600, replace=False)
    sph["p"][odd[:200], 0] = np.float32("nan")
    sph["rad"][odd[200:400]] = np.float32("inf")
    sph["p"][odd[400:], 1] = np.float32("-inf")
    for k, at in enumerate(((-30, 40, 20), (45, 25, -30))):        # two more lights, one of them late in the table
        i = (5, n - 2)[k]
        sph["rad"][i], sph["p"][i], sph["e"][i], sph["refl"][i] = 4.0, at, (9, 8, 7), api.DIFF
    w, h = 40, 24
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, 3, threads=16)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))           # no hierarchy: the fallback form
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.render_pass(1)
        got = {"pixels": ctx.render_pass(2), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        assert ctx.last_kernel == "rt_trace_parity_g"
        _same(got, want)


def test_many_scattered_updates_then_one_launch():
    """Round 6: rt_update_spheres_async sends its records at once and leaves the tables and the hierarchy to the next reader, which builds them
    ONCE (rt_scene.hip refresh_tables) -- 120 scattered spheres moved by 120 calls on two streams, a light switched off and another on; the
    hierarchy read back BEFORE any launch (the diagnostics' reader refreshes too) is a valid tree of the moved scene, the frame the oracle's;
    then more updates and a launch with no reader in between."""
    sph, orig, target = scenes.random_spheres(1024)
    sph = sph.copy()
    w, h, spp = 64, 40, 2
    cam = host.compute_camera(orig, target, w, h)
    rng = np.random.default_rng(8)
    with api.RtContext(w, h, diag=True) as ctx, api.RtContext(8, 8, diag=True) as other:      # (the second context lends its stream)
        side = other.stream
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.render_pass(spp)
        for rnd in range(2):
            who = rng.choice(np.arange(2, len(sph)), 120, replace=False)
            sph["p"][who] += rng.uniform(-3, 3, (120, 3)).astype(np.float32)
            sph["rad"][who[:10]] *= np.float32(1.7)
            sph["e"][who[10]] = (5.0, 4.0, 3.0)                                # a second light appears ...
            sph["e"][who[11]] = (0.0, 0.0, 0.0)
            for k, i in enumerate(who):
                ctx.update_spheres(int(i), sph[int(i):int(i) + 1], side if k % 2 else ctx.stream)
            if rnd == 0:
                assert bvh_check.check_structure(api.as_spheres(sph), bvh_check.read_bvh(ctx)) == []
            ctx.reset_async(ctx.stream)
            ctx.render_async(spp, ctx.stream)
            got = {"pixels": ctx.read_pixels(), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
            _same(got, O.render(sph, cam, w, h, spp, threads=16))
        assert bvh_check.check_structure(api.as_spheres(sph), bvh_check.read_bvh(ctx)) == []


def _two_size_classes(n_small, n_large, seed=3):
    """Dust among objects fifty times its size, a ground sphere and a light (tools/always_list_probe.py)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import always_list_probe
    return always_list_probe.two_classes(n_small, n_large, seed)


@pytest.mark.parametrize("n_small,n_large,by_area", [(700, 300, 1), (3000, 600, 1), (3000, 600, 2), (9000, 1500, 1), (300, 700, 1)])
def test_two_size_classes_both_hang_in_the_tree(n_small, n_large, by_area):
    """Round 6: the cut between the tree and the always-list is never below an eighth of the scene's extent, so the larger class of a scene of
    two size classes -- more than 16 x the median radius, yet nowhere near the scene's size -- hangs in the tree instead of being swept by every
    ray (only the ground sphere stays outside).  Host-shaped and device-shaped trees, LDS and L2 walks; frames, seeds and counters are the
    oracle's, the tables a valid hierarchy."""
    sph, orig, target = _two_size_classes(n_small, n_large)
    w, h, spp = 48, 32, 2
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp, threads=16)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_tree_shape(ctx._h, by_area))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        got = {"pixels": ctx.render_pass(spp), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        assert "_pairs" in ctx.last_kernel
        _same(got, want)
        b = bvh_check.read_bvh(ctx)
        assert b["n_always"] == 1, b["n_always"]                 # the ground; the light (9) and the larger class are in the tree
        assert bvh_check.check_structure(api.as_spheres(sph), b) == []
        # ... and after a device-resident update of some of the larger spheres (the cut is formed again from the mirror)
        sph["p"][-5:] += np.float32([1.0, 0.0, -1.0])
        ctx.update_spheres(len(sph) - 5, sph[-5:], ctx.stream)
        ctx.reset_async(ctx.stream)
        ctx.render_async(spp, ctx.stream)
        got = {"pixels": ctx.read_pixels(), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
        _same(got, O.render(sph, cam, w, h, spp, threads=16))
        assert bvh_check.check_structure(api.as_spheres(sph), bvh_check.read_bvh(ctx)) == []


def test_plain_sweep_through_the_scalar_cache_sees_device_resident_updates():
    """The table rt_trace_parity_g reads through the scalar cache is rewritten on the stream between launches (rt_update_spheres_async ->
    rt_build_tables_kernel): every launch must see the records as they are then -- five frames of spheres moving and a light changing,
    queued without a host wait in between, each against the oracle."""
    n = 9740
    sph, orig, target = _many_spheres(n, seed=5)
    w, h, spp = 40, 24, 2
    cam = host.compute_camera(orig, target, w, h)
    rng = np.random.default_rng(5)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 0, 0))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        for f in range(5):
            who = np.sort(rng.choice(np.arange(2, n), 40, replace=False))
            sph["p"][who] += rng.uniform(-2, 2, (40, 3)).astype(np.float32)
            sph["rad"][who] *= np.float32(1.5)
            sph["p"][1] += np.float32([3.0, -1.0, 2.0])                        # the light
            for i in who:
                ctx.update_spheres(int(i), sph[int(i):int(i) + 1], ctx.stream)
            ctx.update_spheres(1, sph[1:2], ctx.stream)
            ctx.reset_async(ctx.stream)
            ctx.render_async(spp, ctx.stream)
            got = {"pixels": ctx.read_pixels(), "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}
            assert ctx.last_kernel == "rt_trace_parity_g"
            _same(got, O.render(sph, cam, w, h, spp, threads=16))


def test_the_scene_size_limit():
    too_many = np.zeros(262144 + 1, api.SPHERE_DT)
    with api.RtContext(32, 32) as ctx:
        with pytest.raises(api.RtError):
            ctx.set_scene(too_many)
