"""The oracle against (a) the committed golden vectors, which are outputs of the reference
itself (tests/golden/make_golden.py), (b) the reference compiled in place where it exists,
(c) the known values SURVEY.md 8c records (camera hex, first rand() outputs, ray statistics).
CPU only."""
import glob
import os

import numpy as np
import pytest

import _oracle as O


def _load(path):
    z = np.load(path)
    sph = z["spheres"].view(O.SPHERE_DT)
    return z, sph


def _cases(golden_dir):
    return sorted(p for p in glob.glob(os.path.join(golden_dir, "*.npz"))
                  if not p.endswith("host_pins.npz"))


def test_golden_present(golden_dir):
    assert len(_cases(golden_dir)) >= 10


@pytest.mark.parametrize("name", [os.path.basename(p) for p in
                                  _cases(os.path.join(os.path.dirname(__file__), "golden"))])
def test_oracle_reproduces_reference_outputs(golden_dir, name):
    z, sph = _load(os.path.join(golden_dir, name))
    w, h, spp = int(z["w"]), int(z["h"]), int(z["spp"])
    out = O.render(sph, z["camera"], w, h, spp)
    assert np.array_equal(out["pixels"], z["pixels"])
    assert np.array_equal(out["colors"].view(np.uint32), z["colors"].view(np.uint32))
    assert O.fnv(out["seeds"]) == str(z["fnv_seeds"])
    assert O.fnv(out["pixels"]) == str(z["fnv_pixels"])


def test_thread_split_is_bit_invisible(golden_dir):
    z, sph = _load(os.path.join(golden_dir, "demo_200x120_3spp.npz"))
    a = O.render(sph, z["camera"], 200, 120, 3, threads=1)
    b = O.render(sph, z["camera"], 200, 120, 3, threads=7)
    for k in ("pixels", "seeds"):
        assert np.array_equal(a[k], b[k])
    assert a["stats"] == b["stats"]


def test_progressive_equals_one_shot(golden_dir):
    """spp passes in two calls (state carried in colours/seeds) == one call."""
    z, sph = _load(os.path.join(golden_dir, "demo_128x96_16spp.npz"))
    a = O.render(sph, z["camera"], 128, 96, 5)
    b = O.render(sph, z["camera"], 128, 96, 11, first_sample=5, seeds_in=a["seeds"],
                 colors_in=a["colors"])
    assert np.array_equal(b["pixels"], z["pixels"])


def test_seed_stream_pins(golden_dir):
    pins = np.load(os.path.join(golden_dir, "host_pins.npz"))
    raw = np.zeros(8, np.uint32)
    O.oracle().orc_glibc_rand_stream(raw.ctypes.data_as(O.C.c_void_p), 8)
    # first outputs of the never-seeded glibc rand()
    assert list(raw[:4]) == [1804289383, 846930886, 1681692777, 1714636915]
    assert np.array_equal(O.seeds(64, 64), pins["seeds_first_8192"])
    for (w, h) in [(256, 256), (800, 600), (1920, 1080)]:
        assert O.fnv(O.seeds(w, h)) == str(pins[f"fnv_seeds_{w}x{h}"])
    assert O.seeds(1920, 1080).min() >= 2


def test_seed_stream_equals_host_libc():
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    mine = np.zeros(100000, np.uint32)
    O.oracle().orc_glibc_rand_stream(mine.ctypes.data_as(O.C.c_void_p), len(mine))
    theirs = np.array([libc.rand() for _ in range(len(mine))], np.uint32)
    assert np.array_equal(mine, theirs)


def test_camera_basis_pins(golden_dir):
    pins = np.load(os.path.join(golden_dir, "host_pins.npz"))
    # SURVEY a15 golden hex for orig (20,100,120) target (0,25,0)
    want = {
        (256, 256): "be0f4d08 bf065838 bf56f38c 3f465387 00000000 be0437af bd8ac56b 3f2b26af bed02822",
        (1920, 1080): "be0f4d08 bf065838 bf56f38c 3fb04a3e 00000000 be6b0da8 bd8ac56b 3f2b26af bed02820",
    }
    for (w, h), hexes in want.items():
        cam = O.camera(O.DEMO_ORIG, O.DEMO_TARGET, w, h)
        assert " ".join("%08x" % v for v in cam.view(np.uint32)[6:]) == hexes
    for (w, h) in [(256, 256), (800, 600), (1920, 1080), (3840, 2160)]:
        cam = O.camera(O.DEMO_ORIG, O.DEMO_TARGET, w, h)
        assert np.array_equal(cam.view(np.uint32), pins[f"camera_{w}x{h}"].view(np.uint32))


def test_ray_statistics_match_survey():
    """Demo 256^2 -> 2.106 closest-hit, 0.337 shadow, 14.52 tests, 5.32 draws per sample."""
    cam = O.camera(O.DEMO_ORIG, O.DEMO_TARGET, 256, 256)
    st = O.render(O.demo_spheres(), cam, 256, 256, 8)["stats"]
    n = st["samples"]
    assert n == 256 * 256 * 8
    assert abs(st["closest_calls"] / n - 2.106) < 0.01
    assert abs(st["shadow_calls"] / n - 0.337) < 0.01
    assert abs(st["sphere_tests"] / n - 14.52) < 0.1
    assert abs(st["rng_draws"] / n - 5.32) < 0.03


def test_demo_scene_literal_equals_reference_scene(golden_dir):
    z, sph = _load(os.path.join(golden_dir, "c1_demo_256x256_1spp.npz"))
    assert sph.tobytes() == O.demo_spheres().tobytes()


@pytest.mark.skipif(not O.ref_tree_available(), reason="reference build exists only in the build container")
def test_oracle_equals_reference_build_both_backends():
    """Restatement vs the reference kernel compiled in place, on a scene not in the fixtures'
    sizes; back-end 1 (host libm) and back-end 0 (restated libm) must both be bit-equal."""
    sph, orig, target = O.ref_read_scene(os.path.join(O.REF_ROOT, "SimpleRT/Scene/cornell.scn"))
    w, h, spp = 80, 56, 3
    cam = O.camera(orig, target, w, h)
    want = O.ref_render(sph, cam, w, h, spp)
    for backend in (0, 1):
        got = O.render(sph, cam, w, h, spp, backend=backend)
        assert np.array_equal(got["pixels"], want["pixels"])
        assert np.array_equal(got["colors"].view(np.uint32), want["colors"].view(np.uint32))
        assert np.array_equal(got["seeds"], want["seeds"])


@pytest.mark.skipif(not O.ref_available(), reason="reference build exists only where oracle/_ref was built")
def test_threaded_reference_leg_equals_the_sequential_launches():
    """bench.py's cpu_baseline (kind "reference") runs the reference kernel on several host threads,
    each doing all passes over its slice of work-items; the buffers equal launch-by-launch order."""
    sph = O.demo_spheres()
    w, h, spp = 96, 50, 5
    cam = O.camera((20.0, 100.0, 120.0), (0.0, 25.0, 0.0), w, h)
    want = O.ref_render(sph, cam, w, h, spp)
    got = O.ref_render_mt(sph, cam, w, h, spp, 5)
    for key in ("pixels", "colors", "seeds"):
        assert np.array_equal(got[key].view(np.uint32), want[key].view(np.uint32))
