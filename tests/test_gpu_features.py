"""Round-2 rows of the scope table, each against the oracle through the C ABI: the multi-device context
(shards -> one gather -> de-interleave kernel), device-resident scene updates (8f-4),
rt_read_pixels, the cached one-shot rt_render."""
import ctypes as C
import time

import numpy as np
import pytest

import _oracle as O
from raytracing_simple_amd import api, host, scenes

pytestmark = pytest.mark.gpu


def _state(ctx, px):
    return {"pixels": px, "colors": ctx.read_colors(), "seeds": ctx.read_seeds(), "stats": ctx.stats()}


def _assert_same(got, want):
    assert np.array_equal(got["pixels"], want["pixels"])
    assert np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))
    assert np.array_equal(got["seeds"], want["seeds"])
    g, o = got["stats"], want["stats"]
    assert (g["samples"], g["closest_rays"], g["shadow_rays"], g["sphere_tests"], g["rng_draws"]) == \
           (o["samples"], o["closest_calls"], o["shadow_calls"], o["sphere_tests"], o["rng_draws"])


# ---- multi-device context ----------------------------------------------------------------------------
@pytest.mark.parametrize("devices,tile_rows,w,h", [
    ([0], 8, 160, 96),                  # ngpus = 1: nothing to move -- no communicator, no gather: the one shard renders into the frame
    ([0, 0], 8, 160, 96),               # one-GPU rehearsal: shards -> D2D stand-in for the recv -> de-interleave kernel
    ([0, 0, 0], 16, 97, 61),            # ragged: w % 4 != 0 (scalar de-interleave), short last tile, uneven tile counts
    ([0] * 8, 8, 256, 200),
    ([0] * 8, 8, 64, 24),               # more shards than row tiles: five shards own nothing
])
def test_multi_device_context_equals_the_oracle(devices, tile_rows, w, h):
    sph, orig, target = scenes.demo_plus(16)
    cam = host.compute_camera(orig, target, w, h)
    spp = 5
    want = O.render(sph, cam, w, h, spp)
    with api.RtContext(w, h, devices=devices, tile_rows=tile_rows) as ctx:
        assert ctx.shard_count == len(devices) and ctx.local_rows == h
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        _assert_same(_state(ctx, ctx.render_pass(spp)), want)
        # progressive: the same frame in three launches after a reset, the middle one without pixel stores
        ctx.reset()
        ctx.render_pass(2)
        ctx.set_pixel_write(False)
        ctx.render_pass(2)
        assert np.array_equal(ctx.read_pixels(), O.render(sph, cam, w, h, 4)["pixels"])      # packed on demand, then gathered
        ctx.set_pixel_write(True)
        _assert_same(_state(ctx, ctx.render_pass(1)), want)
        # asynchronous reset + fast mode round trip leave the parity result untouched
        ctx.reset_async()
        ctx.set_mode(api.RT_MODE_FAST)
        fast = ctx.render_pass(spp)
        assert host.psnr(fast, want["pixels"]) >= 45.0
        ctx.reset()
        ctx.set_mode(api.RT_MODE_PARITY)
        _assert_same(_state(ctx, ctx.render_pass(spp)), want)


@pytest.mark.parametrize("w,h,spp", [(1920, 1080, 64), (3840, 2160, 256)], ids=["C2", "C4"])
def test_multi_device_context_at_full_size_equals_the_single_device_frame(w, h, spp):
    """C2 (1080p, 64 spp) and C4 (2160p, 256 spp: BASELINE's scaling configuration) on 8 emulated shards = the unsharded frame,
    bit for bit, counters included."""
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    with api.RtContext(w, h) as one:
        one.set_scene(sph); one.set_camera(cam)
        want = _state(one, one.render_pass(spp))
    with api.RtContext(w, h, devices=[0] * 8) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        got = _state(ctx, ctx.render_pass(spp))
    assert np.array_equal(got["pixels"], want["pixels"])
    assert np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))
    assert np.array_equal(got["seeds"], want["seeds"])
    assert {k: v for k, v in got["stats"].items() if k not in ("launches", "last_kernel_ms")} == \
           {k: v for k, v in want["stats"].items() if k not in ("launches", "last_kernel_ms")}


def test_multi_device_argument_checks():
    lib = api.load_library()
    h_ = C.c_void_p()
    assert lib.rt_create_multi(C.byref(h_), 64, 64, 0) == -1
    assert lib.rt_create_multi(C.byref(h_), 64, 64, 99) == -1                 # more devices than the box has
    arr = (C.c_int * 2)(0, 7)
    assert lib.rt_create_multi_on(C.byref(h_), 64, 64, arr, 2, 8) == -1
    arr = (C.c_int * 2)(0, 0)
    assert lib.rt_create_multi_on(C.byref(h_), 64, 64, arr, 2, 12) == -1      # tile_rows % 8
    with api.RtContext(32, 32, devices=[0, 0]) as ctx:
        with pytest.raises(api.RtError):
            ctx.set_pixel_buffer(1234, 32 * 32)                              # a multi-device context owns its frame buffer


@pytest.mark.parametrize("w,h,n,tr", [(1920, 1080, 8, 8), (97, 61, 3, 16), (64, 24, 8, 8), (640, 360, 2, 8), (33, 17, 1, 8)])
def test_deinterleave_kernel_equals_the_row_permutation(w, h, n, tr):
    import torch
    from raytracing_simple_amd import dist as rdist
    dev = torch.device("cuda", 0)
    pad = rdist.max_local_rows(h, n, tr)
    rng = np.random.default_rng(w * h + n)
    parts = [rng.integers(0, 2 ** 31, (pad, w), dtype=np.int64).astype(np.int32) for _ in range(n)]
    stacked = torch.tensor(np.stack(parts), device=dev)
    full = torch.zeros((h, w), dtype=torch.int32, device=dev)
    api.deinterleave_rows(full.data_ptr(), stacked.data_ptr(), w, h, n, tr, pad, device=0,
                          stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = rdist.assemble_numpy([p.view(np.uint32) for p in parts], h, w, n, tr)
    assert np.array_equal(full.cpu().numpy().view(np.uint32).reshape(-1), want)


# ---- device-resident scene updates ----------------------------------------------------------------------
def test_moving_spheres_and_camera_equal_the_oracle_frame_by_frame():
    """Eight frames of an animation: two spheres move (one of them the light: the light list and its 4 pi r^2
    are rebuilt on the device), a third changes material and starts to emit half-way (the light list grows),
    the camera orbits.  Every frame restarts from the default seed stream and must equal the oracle."""
    w, h, spp = 128, 80, 3
    sph, orig, target = scenes.demo_plus(16)
    sph = sph.copy()
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph)
        for f in range(8):
            sph["p"][3] += np.float32([1.5, 0.25, -0.75])                      # a diffuse sphere
            lights = [i for i in range(len(sph)) if sph["e"][i][0] != 0 or sph["e"][i][2] != 0]
            sph["p"][lights[0]] += np.float32([-0.5, 0.5, 0.25])               # the light itself
            sph["rad"][lights[0]] *= np.float32(1.03)
            ctx.update_spheres(3, sph[3:4], ctx.stream)
            ctx.update_spheres(lights[0], sph[lights[0]:lights[0] + 1], ctx.stream)
            if f == 4:
                sph["refl"][7] = api.SPEC
                sph["e"][9] = (3.0, 2.0, 1.0)                                  # a second light appears
                ctx.update_spheres(7, sph[7:10], ctx.stream)
            o = np.float32(orig) + np.float32([3.0 * f, 1.0 * f, -2.0 * f])
            cam = host.compute_camera(tuple(float(v) for v in o), target, w, h)
            ctx.set_camera(cam)
            ctx.reset_async(ctx.stream)
            ctx.render_async(spp, ctx.stream)
            got = _state(ctx, ctx.read_pixels())
            _assert_same(got, O.render(sph, cam, w, h, spp))


def test_update_spheres_argument_checks_and_refused_scene_keeps_the_old_one():
    w, h = 48, 32
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    want = O.render(sph, cam, w, h, 2)["pixels"]
    with api.RtContext(w, h) as ctx:
        with pytest.raises(api.RtError):
            ctx.update_spheres(0, sph[:1])                                     # no scene yet
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        with pytest.raises(api.RtError):
            ctx.update_spheres(5, sph[:2])                                     # runs past the end
        too_many = np.zeros(262144 + 1, api.SPHERE_DT)                         # RT_MAX_SPHERES + 1
        with pytest.raises(api.RtError):
            ctx.set_scene(too_many)
        assert np.array_equal(ctx.render_pass(2), want)                        # the Demo scene is still in place


def test_thousands_of_lights_do_not_need_lds():
    """6000 emitters: the light list alone (32 B each) is beyond LDS; every diffuse hit samples all of them."""
    rng = np.random.default_rng(11)
    n = 6000
    sph = np.zeros(n, api.SPHERE_DT)
    sph["rad"] = rng.uniform(0.2, 0.6, n).astype(np.float32)
    sph["p"] = np.stack([rng.uniform(-60, 60, n), rng.uniform(5, 40, n), rng.uniform(-60, 60, n)], 1).astype(np.float32)
    sph["e"] = rng.uniform(0.5, 3.0, (n, 3)).astype(np.float32)
    sph["rad"][0], sph["p"][0], sph["e"][0], sph["c"][0] = 1000.0, (0, -1000, 0), (0, 0, 0), (.7, .7, .7)
    w, h, spp = 16, 12, 1
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    want = O.render(sph, cam, w, h, spp, threads=16)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        assert np.array_equal(ctx.render_pass(spp), want["pixels"])
        assert ctx.last_kernel.endswith("_g")
        st = ctx.stats()
        assert (st["shadow_rays"], st["sphere_tests"], st["rng_draws"]) == (want["stats"]["shadow_calls"], want["stats"]["sphere_tests"], want["stats"]["rng_draws"])


def test_scene_replaced_between_frames_without_a_device_wide_wait():
    """rt_set_scene while another context keeps the GPU busy: ordered on the context's own work only."""
    w, h = 96, 64
    a, b = scenes.demo_plus(16), scenes.random_spheres(96)
    with api.RtContext(w, h) as ctx, api.RtContext(640, 360) as busy:
        busy.set_scene(host.demo_scene())
        busy.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 640, 360))
        for rnd in range(6):
            busy.render_async(16, busy.stream)
            sph, orig, target = (a, b)[rnd % 2]
            cam = host.compute_camera(orig, target, w, h)
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            ctx.reset_async(ctx.stream)
            ctx.render_async(2, ctx.stream)
            assert np.array_equal(ctx.read_pixels(), O.render(sph, cam, w, h, 2)["pixels"])


# ---- rt_read_pixels --------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [api.RT_MODE_PARITY, api.RT_MODE_FAST])
def test_read_pixels_packs_the_frame_the_skipped_launches_left_behind(mode):
    w, h = 120, 72
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    with api.RtContext(w, h) as ref, api.RtContext(w, h, rank=1, nranks=3) as shard, api.RtContext(w, h) as ctx:
        for c in (ref, shard, ctx):
            c.set_scene(sph); c.set_camera(cam); c.set_mode(mode)
        want = ref.render_pass(7)                                              # pixel store on: the kernel's own toInt
        ctx.set_pixel_write(False)
        ctx.render_pass(3)
        ctx.render_async(4, ctx.stream)
        assert np.array_equal(ctx.read_pixels(), want)
        assert np.array_equal(ctx.read_pixels(), want)                         # idempotent
        shard.set_pixel_write(False)
        shard.render_pass(7)
        rows = shard.local_row_map()
        assert np.array_equal(shard.read_pixels().reshape(-1, w), want.reshape(h, w)[rows])
    if mode == api.RT_MODE_PARITY:
        assert np.array_equal(want, O.render(sph, cam, w, h, 7)["pixels"])


# ---- the headline call, repeated --------------------------------------------------------------------------
def test_rt_render_repeated_calls_reuse_device_state_and_stay_bit_exact():
    lib = api.load_library()
    lib.rt_release_cache()
    for rnd in range(3):
        for (w, h, spp, maker) in ((96, 64, 3, lambda: scenes.demo_plus(16)), (64, 48, 2, lambda: scenes.random_spheres(96)),
                                   (96, 64, 1, lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET)),
                                   (33, 17, 4, lambda: scenes.mirror_box(64)), (7, 3, 2, lambda: scenes.demo_plus(16)),
                                   (40, 24, 0, lambda: scenes.demo_plus(16))):
            sph, orig, target = maker()
            cam = host.compute_camera(orig, target, w, h)
            assert np.array_equal(api.render(sph, cam, w, h, spp), O.render(sph, cam, w, h, spp)["pixels"]), (rnd, w, h)
    # a failing call must not poison the cache
    with pytest.raises(api.RtError):
        api.render(np.zeros(262144 + 1, api.SPHERE_DT), host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 96, 64), 96, 64, 1)
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, 96, 64)
    assert np.array_equal(api.render(sph, cam, 96, 64, 2), O.render(sph, cam, 96, 64, 2)["pixels"])
    lib.rt_release_cache()
    assert np.array_equal(api.render(sph, cam, 96, 64, 2), O.render(sph, cam, 96, 64, 2)["pixels"])


def test_rt_render_second_call_costs_little_more_than_its_kernel():
    """at 1080p x 64 spp the second and later calls take <= kernel + 1.5 ms of wall time."""
    w, h, spp = 1920, 1080, 64
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        first = ctx.render_pass(spp)
        kernel_ms = ctx.stats()["last_kernel_ms"]
    api.render(sph, cam, w, h, spp)
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        px = api.render(sph, cam, w, h, spp)
        times.append((time.perf_counter() - t0) * 1e3)
    assert np.array_equal(px, first)
    print("rt_render at 1080p x 64 spp: calls", ["%.2f" % t for t in times], "ms; kernel", "%.2f" % kernel_ms, "ms")
    assert sorted(times)[len(times) // 2] <= kernel_ms + 1.5


# ---- heavy tiles first -------------------------------------------------------------------------------------
def test_a_first_long_frame_prices_its_tiles_with_four_of_its_own_passes():
    """A launch of 24 passes or more that has no tile costs to go by -- the first frame of a scene -- renders 4 of its passes first
    and the rest heavy first (rt_launch.hip launch_priced): two launches instead of one, the same frame bit for bit; from the
    second frame on one launch.  Shorter frames are not split."""
    w, h = 160, 96
    sph, orig, target = scenes.demo_plus(16)
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        want = O.render(sph, cam, w, h, 24)
        launches = []
        for _ in range(4):
            ctx.reset()
            _assert_same(_state(ctx, ctx.render_pass(24)), want)
            launches.append(ctx.stats()["launches"])
        assert launches == [2, 1, 1, 1]
        ctx.set_scene(scenes.demo_plus(12)[0])                                 # another scene: priced again
        ctx.reset(); ctx.render_pass(30)
        assert ctx.stats()["launches"] == 2
        ctx.set_scene(scenes.demo_plus(13)[0])                                 # ... but not for a short frame
        ctx.reset(); ctx.render_pass(20)
        assert ctx.stats()["launches"] == 1
        # progressive launches split the same way and stay the oracle's
        ctx.set_scene(sph)
        ctx.reset(); ctx.render_pass(40); ctx.render_pass(24)
        _assert_same(_state(ctx, ctx.read_pixels()), O.render(sph, cam, w, h, 64))


def test_small_scenes_measure_cooperative_any_hit_against_plain_on_their_own_launches():
    """Scenes of 4 to 11 spheres (no hierarchy, below the cooperative instances' threshold): the sweep WITH the wave-ballot any-hit sharing and
    WITHOUT it are timed on the host's own launches as they come (rt_launch.hip launch_small) -- never split, scheduled like any other -- and the
    faster instance renders the rest.  A host that renders whole frames: first frame by the threshold's pick (it is never part of the measurement),
    second frame cooperative, third plain, verdict for the fourth; a host that queues a pass per call: one launch and 16 passes per form.  Frames,
    colour plane, seeds and counters are the oracle's throughout; a threshold set by hand switches the measurement off."""
    w, h = 200, 120
    for maker in (lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), lambda: scenes.demo_plus(10)):
        sph, orig, target = maker()
        cam = host.compute_camera(orig, target, w, h)
        want = O.render(sph, cam, w, h, 40)
        with api.RtContext(w, h) as ctx:
            ctx.set_scene(sph); ctx.set_camera(cam)
            kernels, launches = [], []
            for _ in range(5):
                ctx.reset()
                _assert_same(_state(ctx, ctx.render_pass(40)), want)
                kernels.append(ctx.last_kernel)
                launches.append(ctx.stats()["launches"])
            assert launches == [2, 1, 1, 1, 1]                                   # (the first frame prices its tiles; no frame is ever split for the measurement)
            assert kernels[:3] == ["rt_trace_parity_w1", "rt_trace_parity_coop_w1", "rt_trace_parity_w1"]
            picked = kernels[3]
            assert picked in ("rt_trace_parity_w1", "rt_trace_parity_coop_w1") and kernels[4] == picked
            ctx.update_spheres(1, api.as_spheres(sph)[1:2])                      # a device-resident update keeps the verdict (the sphere count cannot change)
            ctx.reset(); ctx.render_pass(40)
            assert ctx.stats()["launches"] == 1 and ctx.last_kernel == picked
            fewer = np.concatenate([api.as_spheres(sph)[:4], api.as_spheres(sph)[5:]])          # (one sphere less; the light stays)
            ctx.set_scene(fewer)                                                 # another scene: measured again, here through queued one-pass launches
            ctx.reset()
            names = []
            for _ in range(36):
                ctx.render_async(1)
                names.append(ctx.last_kernel)
            assert names[:17] == ["rt_trace_parity_coop_w1"] * 17 and names[17:34] == ["rt_trace_parity_w1"] * 17, names
            assert np.array_equal(ctx.read_pixels(), O.render(fewer, cam, w, h, 36)["pixels"])
        with api.RtContext(w, h, diag=True) as ctx:                               # a threshold set by hand decides alone
            ctx._check(ctx._lib.rt_debug_set_coop_min(ctx._h, 12))
            ctx.set_scene(sph); ctx.set_camera(cam)
            for _ in range(3):
                ctx.reset()
                _assert_same(_state(ctx, ctx.render_pass(40)), want)
                assert ctx.last_kernel == "rt_trace_parity_w1"


def test_short_launches_build_their_own_tile_order_from_a_window_of_costs():
    """The reference's regime is a pass per call: launches too short to price tiles one by one.  While the order is missing, short launches ADD their
    per-tile costs up (the kernel's epilogue adds instead of stores) and the launch that finds 16 passes' worth sorts from them; from then on short
    launches walk that order and leave the costs alone; a moved camera starts the window again.  Frames are the oracle's throughout."""
    lib = api.load_library(diag=True)
    w, h = 200, 120
    sph, orig, target = scenes.demo_plus(16)
    cam = host.compute_camera(orig, target, w, h)

    def order_state(ctx):
        n, valid = C.c_uint32(), C.c_int()
        order, cost = np.zeros(4096, np.uint32), np.zeros(4096, np.uint32)
        api._check(lib.rt_debug_read_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), 4096, C.byref(n), C.byref(valid)), lib)
        return order[: n.value], cost[: n.value], bool(valid.value)

    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        sums = []
        for k in range(16):                                                    # 16 launches of one pass: the window fills, nothing is sorted yet
            ctx.render_pass(1)
            assert not order_state(ctx)[2]
            sums.append(int(order_state(ctx)[1].astype(np.uint64).sum()))
        assert all(b > a for a, b in zip(sums, sums[1:]))                      # the costs add up launch by launch
        window = order_state(ctx)[1].copy()
        ctx.render_pass(1)                                                     # the 17th finds 16 passes' worth: sorted, and walked from here on
        order, cost, valid = order_state(ctx)
        assert valid and sorted(order.tolist()) == list(range(len(order))) and np.array_equal(cost, window)
        capped = np.minimum(window, 0x1FFFFF).astype(np.uint64)
        cls = 1023 - (capped * 1023 // int(capped.max())).astype(np.int64)
        assert np.all(np.diff(cls[order]) >= 0)                                # most expensive class first
        for _ in range(7):
            ctx.render_pass(1)
        assert np.array_equal(order_state(ctx)[1], window)                     # (a valid order: short launches leave the costs alone)
        assert np.array_equal(ctx.read_pixels(), O.render(sph, cam, w, h, 24)["pixels"])
        cam2 = host.compute_camera((30.0, 90.0, 110.0), target, w, h)
        ctx.set_camera(cam2)                                                   # stale: the window starts again at the new camera, the old order stays in use meanwhile
        ctx.reset()
        for k in range(17):
            ctx.render_pass(1)
            assert order_state(ctx)[2]
            if k == 0:
                assert int(order_state(ctx)[1].astype(np.uint64).sum()) < int(window.astype(np.uint64).sum()) // 4     # (one pass' worth: the window was restarted)
        assert not np.array_equal(order_state(ctx)[0], order)                  # sorted again from the new window
        assert np.array_equal(ctx.read_pixels(), O.render(sph, cam2, w, h, 17)["pixels"])


def test_the_order_of_a_priced_first_frame_is_sorted_once_more_from_the_whole_frame():
    """The first frame's order comes from 4 passes' worth of costs; the frame itself leaves the costs of all its passes, and the second
    frame sorts from those -- once: the third frame walks the second's order.  Frames are the oracle's throughout."""
    lib = api.load_library(diag=True)
    w, h, spp = 320, 200, 32
    sph, orig, target = scenes.demo_plus(16)
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)

    def order_of(ctx):
        n, valid = C.c_uint32(), C.c_int()
        order = np.zeros(8192, np.uint32)
        api._check(lib.rt_debug_read_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), None, 8192, C.byref(n), C.byref(valid)), lib)
        assert valid.value
        return order[: n.value].copy()

    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        orders = []
        for k in range(4):
            ctx.reset()
            _assert_same(_state(ctx, ctx.render_pass(spp)), want)
            orders.append(order_of(ctx))
            assert ctx.stats()["launches"] == (2 if k == 0 else 1)
        assert not np.array_equal(orders[0], orders[1])                        # sorted again: from the first frame's 28 remaining passes
        assert np.array_equal(orders[1], orders[2]) and np.array_equal(orders[2], orders[3])


def test_heavy_first_tile_order_is_a_permutation_sorted_by_cost_and_changes_no_bit():
    """Long launches of one scene and camera walk the tiles in descending order of the cost the launch before measured.
    Scheduling only: pixels, colour plane, seeds and counters equal the oracle either way; a new scene drops the order until
    costs exist again; a moved camera or an updated scene sorts it again from the last frame's costs."""
    lib = api.load_library(diag=True)
    w, h, spp = 200, 120, 8
    sph, orig, target = scenes.demo_plus(16)
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)

    def order_state(ctx):
        n, valid = C.c_uint32(), C.c_int()
        order = np.zeros(4096, np.uint32)
        cost = np.zeros(4096, np.uint32)
        api._check(lib.rt_debug_read_tile_order(ctx._h, order.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), 4096,
                                                C.byref(n), C.byref(valid)), lib)
        return order[: n.value], cost[: n.value], bool(valid.value)

    with api.RtContext(w, h, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        _assert_same(_state(ctx, ctx.render_pass(spp)), want)                  # natural order, leaves costs
        order, cost, valid = order_state(ctx)
        n8 = ((w + 7) // 8) * ((h + 7) // 8)                                   # 16 spheres: single-wavefront workgroups, 8x8 tiles
        assert not valid and len(cost) == n8 and cost[:n8].min() > 0
        ctx.reset()
        _assert_same(_state(ctx, ctx.render_pass(spp)), want)                  # heavy first
        order, _, valid = order_state(ctx)
        assert valid and sorted(order.tolist()) == list(range(len(order)))
        capped = np.minimum(cost, 0x1FFFFF).astype(np.uint64)
        cls = 1023 - (capped * 1023 // int(capped.max())).astype(np.int64)                    # the kernel's cost classes
        assert np.all(np.diff(cls[order]) >= 0)                                # most expensive class first
        # short launches under a valid order neither sort nor touch the costs
        costs_before = order_state(ctx)[1].copy()
        ctx.reset(); ctx.render_pass(2); ctx.render_pass(6)
        _assert_same(_state(ctx, ctx.read_pixels()), want)
        assert np.array_equal(order_state(ctx)[1], costs_before)
        # a moved camera keeps the last long frame's costs (they still predict the next frame): the order stays in use, stale, and the NEXT launch --
        # short or long -- sorts it again from them
        cam2 = host.compute_camera((30.0, 90.0, 110.0), target, w, h)
        ctx.set_camera(cam2)
        assert order_state(ctx)[2]
        stale = order_state(ctx)[0].copy()
        ctx.reset(); ctx.render_pass(2)
        assert order_state(ctx)[2] and not np.array_equal(order_state(ctx)[0], stale)
        fresh = order_state(ctx)[0].copy()
        ctx.reset()
        _assert_same(_state(ctx, ctx.render_pass(spp)), O.render(sph, cam2, w, h, spp))      # (sorted already: this long launch walks that order and leaves its own costs)
        assert order_state(ctx)[2] and np.array_equal(order_state(ctx)[0], fresh)
        ctx.set_scene(sph)                                                     # the identical scene: nothing changes
        ctx.reset()
        _assert_same(_state(ctx, ctx.render_pass(spp)), O.render(sph, cam2, w, h, spp))
        assert order_state(ctx)[2]
        ctx.set_scene(scenes.demo_plus(12)[0])                                 # another scene: costs and order dropped
        ctx.reset(); ctx.render_pass(spp)
        assert not order_state(ctx)[2]
        ctx.set_scene(sph)
        ctx.reset(); ctx.render_pass(spp)
        lib.rt_debug_set_tile_order(ctx._h, 0)                                 # knob: natural order again
        ctx.reset()
        _assert_same(_state(ctx, ctx.render_pass(spp)), O.render(sph, cam2, w, h, spp))
    # sharded contexts order their own tiles
    parts = []
    for r in range(3):
        with api.RtContext(w, h, rank=r, nranks=3, diag=True) as ctx:
            ctx.set_scene(sph); ctx.set_camera(cam)
            for _ in range(3):
                ctx.reset(); px = ctx.render_pass(spp)
            assert order_state(ctx)[2]
            parts.append(px)
    from raytracing_simple_amd import dist as rdist
    assert np.array_equal(rdist.assemble_numpy(parts, h, w, 3, 8), want["pixels"])


# ---- convergence tooling -----------------------------------------------------------------------
def test_convergence_tool_checkpoints_are_the_oracles_frames_and_converge(tmp_path):
    """tools/convergence.py: progressive checkpoints at every power of two, PPM per checkpoint, PSNR / RMSE
    against a long render.  The PPMs must be the oracle's frames of those pass counts, and the error must fall."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w, h = 96, 64
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "convergence.py"), "--scene", "demo", "--w", str(w), "--h", str(h),
                          "--max-spp", "16", "--ref-spp", "256", "--out", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert [r["spp"] for r in rows] == [1, 2, 4, 8, 16]
    assert rows[-1]["psnr_parity_vs_ref_db"] > rows[0]["psnr_parity_vs_ref_db"] + 6.0          # 16x the samples: ~12 dB
    assert rows[-1]["rmse_colour_parity"] < rows[0]["rmse_colour_parity"] / 2
    assert all(r["psnr_fast_vs_parity_same_spp_db"] >= 45.0 for r in rows)
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    for spp in (1, 4, 16):
        raw = (tmp_path / ("demo_parity_%dspp.ppm" % spp)).read_bytes()
        head = b"P6\n%d %d\n255\n" % (w, h)
        assert raw.startswith(head)
        rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)[::-1]
        want = O.render(sph, cam, w, h, spp)["pixels"].view(np.uint8).reshape(h, w, 4)[:, :, :3]
        assert np.array_equal(rgb, want), spp



def test_one_kernel_instance_renders_every_shard():
    """ONE decision per multi-device context: the first shard measures hierarchy against plain sweep, the others launch
    the form it launched -- during the probes and after the verdict -- so rt_last_kernel is true for the whole frame
    (a 64-sphere scene sits at the crossover, where shards deciding for themselves would disagree)."""
    sph, orig, target = scenes.random_spheres(64)
    w, h = 128, 96
    cam = host.compute_camera(orig, target, w, h)
    with api.RtContext(w, h, devices=[0, 0, 0], diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_choice_estimate(ctx._h, 0))           # (measured, whatever the surface-area estimate says of this scene)
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        done = 0
        seen = []
        for n in (1, 2, 1, 2, 3, 24, 7):
            px = ctx.render_pass(n)
            done += n
            kernels = [ctx._lib.rt_debug_shard_kernel(ctx._h, r).decode() for r in range(3)]
            assert len(set(kernels)) == 1 and kernels[0] == ctx.last_kernel, kernels
            seen.append(kernels[0])
            assert np.array_equal(px, O.render(sph, cam, w, h, done)["pixels"])
        sweep = "rt_trace_parity_coop_w1"                                      # 64 spheres: small tables, single-wavefront workgroups
        assert seen[:2] == ["rt_trace_parity_pairs"] * 2 and seen[2:4] == [sweep] * 2      # warm + timed, each form
        ch = ctx.scene_choice()
        assert ch["picked"] in ("hierarchy", "sweep")
        assert seen[-1] == ("rt_trace_parity_pairs" if ch["picked"] == "hierarchy" else sweep)
        st = ctx.stats()
        want = O.render(sph, cam, w, h, done)
        assert (st["samples"], st["sphere_tests"], st["rng_draws"]) == (want["stats"]["samples"], want["stats"]["sphere_tests"], want["stats"]["rng_draws"])


def test_asynchronous_frames_through_the_rehearsal_gather_are_not_torn():
    """rt_render_async on a multi-device context with pixel stores on: frame k's rows are copied out of a shard's pixel
    buffer on the ROOT's stream, so the shard's frame k+1 must wait for that copy (a write-after-read hazard
    on the non-blocking path).  40 progressive launches without a reset in between, the assembled frame read at the end
    and at checkpoints, on a size whose frame takes long enough to copy."""
    sph, orig, target = scenes.demo_plus(16)
    w, h = 1024, 768
    cam = host.compute_camera(orig, target, w, h)
    for devices in ([0, 0], [0, 0, 0]):
        with api.RtContext(w, h, devices=devices) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            done = 0
            # Round 4: frame k is gathered on the root's SECOND stream into receive slot k % 2 while the root's render of frame
            # k + 1 goes into the other slot; a slot is rendered into again only once its previous frame has been assembled
            # out of it.  The checkpoints fall on both slots, right after a slot's first use and after many reuses.
            for k in range(40):
                ctx.render_async(1)
                done += 1
                if k in (0, 1, 17, 18, 39):
                    assert np.array_equal(ctx.read_pixels(), O.render(sph, cam, w, h, done)["pixels"]), (devices, k)
            px = ctx.render_pass(2)                                   # a blocking frame behind the asynchronous ones
            assert np.array_equal(px, O.render(sph, cam, w, h, done + 2)["pixels"])


def test_a_failed_gather_leaves_the_context_in_a_defined_error_state():
    """An RCCL failure inside the gather's group (or at ncclGroupEnd) must not leave later calls queueing behind a
    communicator in an unknown state: the context is marked unusable, every later call is refused with RT_ERR_STATE and
    the failure's name, and rt_destroy still works.  The failure is injected (rt_debug_break_gather)."""
    sph, orig, target = scenes.demo_plus(16)
    w, h = 96, 64
    cam = host.compute_camera(orig, target, w, h)
    ctx = api.RtContext(w, h, devices=[0, 0], diag=True)
    ctx.set_scene(sph)
    ctx.set_camera(cam)
    assert np.array_equal(ctx.render_pass(2), O.render(sph, cam, w, h, 2)["pixels"])
    assert ctx._lib.rt_debug_break_gather(ctx._h) == -3                       # RT_ERR_HIP: what the failing gather itself returns
    for call in (lambda: ctx.render_pass(1), lambda: ctx.render_async(1), lambda: ctx.read_pixels(), lambda: ctx.set_scene(sph),
                 lambda: ctx.set_camera(cam), lambda: ctx.reset(), lambda: ctx.set_mode(api.RT_MODE_FAST)):
        with pytest.raises(api.RtError) as e:
            call()
        assert e.value.code == -5 and "unusable" in str(e.value) and "ncclGroupEnd" in str(e.value)
    ctx.close()
    with api.RtContext(w, h, devices=[0, 0]) as again:                          # a new context is unaffected
        again.set_scene(sph)
        again.set_camera(cam)
        assert np.array_equal(again.render_pass(2), O.render(sph, cam, w, h, 2)["pixels"])


def test_consumer_behind_rt_stream_sees_whole_frames_of_a_multi_device_context():
    """The documented interop pattern on a multi-device context: rt_render_async, then a reader of
    rt_device_pixels queued behind rt_stream(ctx) -- no rt_throttle, no blocking call in between.  Since round 4 the frame is
    assembled on the root's gather stream, so THAT is the stream rt_stream must hand out: every copy taken behind it must be a
    whole frame of the pass count it was queued at, also while the next frames are already being rendered and gathered."""
    import torch
    sph, orig, target = scenes.demo_plus(16)
    w, h = 1024, 768
    cam = host.compute_camera(orig, target, w, h)
    want = {k: torch.from_numpy(O.render(sph, cam, w, h, k)["pixels"].view(np.int32).reshape(h, w).copy()) for k in (1, 2, 3, 9, 10)}
    for devices in ([0, 0], [0, 0, 0], [0]):
        with api.RtContext(w, h, devices=devices) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            frame = torch.as_tensor(ctx.device_pixels_array(), device="cuda")
            reader = torch.cuda.ExternalStream(ctx.stream)
            copies = {}
            for k in range(1, 11):
                ctx.render_async(1)
                if k in want:
                    with torch.cuda.stream(reader):
                        copies[k] = frame.clone()               # queued behind rt_stream(ctx): nothing waited for
            torch.cuda.synchronize()
            for k, got in copies.items():
                assert torch.equal(got.cpu(), want[k]), (devices, k)


@pytest.fixture(scope="module")
def rccl_double_so(tmp_path_factory):
    """tests/rccl_double.cpp built once per module into a scratch directory."""
    import os
    import subprocess
    from raytracing_simple_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("rccl_double") / "librccl_double.so")
    subprocess.run([_build.hipcc(), "-O1", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
                    os.path.join(root, "tests", "rccl_double.cpp"), "-o", so], check=True, capture_output=True)
    return so


@pytest.fixture
def rccl_double(rccl_double_so):
    """The double bound in the place of RCCL by the diagnostics library FOR ONE TEST, with a repeated device list taken as distinct
    devices: rt_multi.hip's grouped ncclRecv / ncclSend branch and its failure handling run on this one GPU.  Yields the double's own
    handle (call counts, failure injection); RCCL -- and with it the one-GPU rehearsal's meaning of a repeated device list -- is
    restored when the test ends, so the rehearsal tests of the diagnostics library keep exercising the emulated path."""
    so = rccl_double_so
    lib = api.load_library(diag=True)
    assert lib.rt_debug_set_rccl_library(so.encode(), 1) == 0
    dbl = C.CDLL(so)                                        # the same handle the library binds (dlopen counts references)
    dbl.rccl_double_fail.argtypes = [C.c_char_p, C.c_long]
    dbl.rccl_double_reset()
    yield dbl
    assert lib.rt_debug_set_rccl_library(None, 0) == 0


def _double_counts(dbl):
    out = (C.c_long * 8)()
    dbl.rccl_double_counts(out)
    return dict(zip(("init", "group_start", "send", "recv", "group_end", "abort", "destroy", "copies"), out))


def test_grouped_send_receive_branch_runs_against_the_double(rccl_double):
    """The in-library n > 1 branch as it runs on real links -- ncclCommInitAll, per frame ONE group of n - 1 receives on the
    root's gather stream and one send per other shard on its own stream, the de-interleave behind them -- executed for real,
    against a double that keeps RCCL's pairing and stream ordering.  Frames, colour plane, seeds and counters equal the
    oracle's; asynchronous frames are whole; the double saw exactly the calls the design promises."""
    sph, orig, target = scenes.demo_plus(16)
    w, h, spp = 320, 200, 6
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    for n in (2, 3, 8):
        rccl_double.rccl_double_reset()
        with api.RtContext(w, h, devices=[0] * n, diag=True) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            _assert_same(_state(ctx, ctx.render_pass(spp)), want)
            ctx.reset()
            for k in range(spp):
                ctx.render_async(1)
                if k == 2:
                    assert np.array_equal(ctx.read_pixels(), O.render(sph, cam, w, h, 3)["pixels"])
            assert np.array_equal(ctx.read_pixels(), want["pixels"])
        got = _double_counts(rccl_double)
        frames = 1 + spp
        assert got["init"] == 1 and got["group_start"] == frames and got["group_end"] == frames
        assert got["recv"] == got["send"] == got["copies"] == (n - 1) * frames
        assert got["destroy"] == n and got["abort"] == 0


def test_c4_at_full_size_through_the_double_equals_the_single_device_frame(rccl_double):
    """BASELINE configs[3] -- Demo, 3840x2160, 256 spp, 8 shards -- through the in-library grouped receive / send branch (against the
    double): the assembled frame, the colour plane, the seeds and the counters are the one-device context's, and the double saw one
    group of 7 receives and 7 sends."""
    w, h, spp = 3840, 2160, 256
    sph = host.demo_scene()
    cam = host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, w, h)
    with api.RtContext(w, h) as one:
        one.set_scene(sph); one.set_camera(cam)
        want = _state(one, one.render_pass(spp))
    rccl_double.rccl_double_reset()
    with api.RtContext(w, h, devices=[0] * 8, diag=True) as ctx:
        ctx.set_scene(sph); ctx.set_camera(cam)
        got = _state(ctx, ctx.render_pass(spp))
    assert np.array_equal(got["pixels"], want["pixels"])
    assert np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))
    assert np.array_equal(got["seeds"], want["seeds"])
    assert {k: v for k, v in got["stats"].items() if k not in ("launches", "last_kernel_ms")} == \
           {k: v for k, v in want["stats"].items() if k not in ("launches", "last_kernel_ms")}
    calls = _double_counts(rccl_double)
    assert calls["init"] == 1 and calls["group_start"] == calls["group_end"] == 1 and calls["recv"] == calls["send"] == 7 and calls["destroy"] == 8


@pytest.mark.parametrize("where,kth", [("ncclSend", 2), ("ncclRecv", 4), ("ncclGroupEnd", 2), ("ncclGroupStart", 3)])
def test_a_failing_rccl_call_breaks_the_context_and_teardown_returns(rccl_double, where, kth):
    """mark_broken for real: the k-th call of one RCCL function fails -- inside the group, at its end, at its
    start.  The frame before it is intact, the failing call returns RT_ERR_HIP naming the function, the communicators are aborted
    (not destroyed), every later call is refused with RT_ERR_STATE, rt_destroy returns, and a new context works."""
    sph, orig, target = scenes.demo_plus(16)
    w, h = 96, 64
    cam = host.compute_camera(orig, target, w, h)
    rccl_double.rccl_double_reset()
    import torch
    with api.RtContext(w, h, devices=[0, 0, 0], diag=True) as first:        # (a first context pays the process-wide allocations: not part of the baseline)
        first.set_scene(sph); first.set_camera(cam); first.render_pass(1)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    rccl_double.rccl_double_reset()
    ctx = api.RtContext(w, h, devices=[0, 0, 0], diag=True)
    ctx.set_scene(sph)
    ctx.set_camera(cam)
    assert np.array_equal(ctx.render_pass(1), O.render(sph, cam, w, h, 1)["pixels"])
    assert torch.cuda.mem_get_info(0)[0] < free_before                          # (the context holds device memory now)
    assert rccl_double.rccl_double_fail(where.encode(), kth) == 0             # the kth call from now (a frame = 1 start, 2 receives, 2 sends, 1 end)
    with pytest.raises(api.RtError) as e:
        for _ in range(3):
            ctx.render_pass(1)
    assert e.value.code == -3 and where in str(e.value) and "unusable" in str(e.value)
    for call in (lambda: ctx.render_pass(1), lambda: ctx.render_async(1), lambda: ctx.read_pixels(), lambda: ctx.set_camera(cam), lambda: ctx.reset()):
        with pytest.raises(api.RtError) as e2:
            call()
        assert e2.value.code == -5 and where in str(e2.value)
    t0 = time.time()
    ctx.close()
    assert time.time() - t0 < 5.0
    # the double's transfers are done (or were never posted): the broken context's streams drain, so its teardown frees what it held --
    # device memory is back at the baseline (a host that destroys and re-creates after a failure must not lose memory per failure)
    assert torch.cuda.mem_get_info(0)[0] >= free_before - (1 << 20), (free_before, torch.cuda.mem_get_info(0)[0])
    got = _double_counts(rccl_double)
    assert got["abort"] == 3 and got["destroy"] == 0
    rccl_double.rccl_double_reset()
    with api.RtContext(w, h, devices=[0, 0], diag=True) as again:
        again.set_scene(sph)
        again.set_camera(cam)
        assert np.array_equal(again.render_pass(2), O.render(sph, cam, w, h, 2)["pixels"])


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="the RCCL send / receive across distinct devices needs at least two GPUs")
def test_multi_device_context_on_distinct_devices_equals_the_one_device_frame():
    """The REAL n > 1 path (ncclCommInitAll over distinct devices, the grouped ncclSend / ncclRecv, the de-interleave on
    the root): never run so far -- every box this suite has seen had one GPU.  On a multi-GPU node the frame, the
    colour plane, the seeds and the counters must equal the one-device context's and the oracle's."""
    n = min(_device_count(), 8)
    sph, orig, target = scenes.demo_plus(16)
    w, h, spp = 320, 200, 6
    cam = host.compute_camera(orig, target, w, h)
    want = O.render(sph, cam, w, h, spp)
    for devices in ([0, 1], list(range(n))):
        with api.RtContext(w, h, devices=devices) as ctx:
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            _assert_same(_state(ctx, ctx.render_pass(spp)), want)
            ctx.reset()
            for _ in range(spp):
                ctx.render_async(1)
            assert np.array_equal(ctx.read_pixels(), want["pixels"])
        with api.RtContext(w, h, devices=devices[::-1]) as ctx:                 # the root need not be device 0
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            assert np.array_equal(ctx.render_pass(spp), want["pixels"])


def test_mixed_device_lists_are_refused():
    with pytest.raises(api.RtError) as e:
        api.RtContext(64, 48, devices=[0, 1, 0])
    assert e.value.code == -1 and "mixed" in str(e.value)


