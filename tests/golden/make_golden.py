#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF: the reference's kernel source and
host sources compiled in place as host C++ (oracle/_ref/libref.so, recipe in
oracle/Makefile + oracle/ref_shim.cpp).  Runs only in the build container, where
/root/reference exists.  Each fixture is data: the kernel's inputs (sphere array as the
reference's loader hands it over -- including readScene's doubling --, camera, size, spp)
and the reference's outputs (pixels; FNV-1a-64 of the colour plane and of the final seeds;
the colour plane itself for the small cases).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SCN = os.path.join(O.REF_ROOT, "SimpleRT", "Scene")

CASES = [
    # name, scene, w, h, spp   (C1 = BASELINE.json configs[0])
    ("c1_demo_256x256_1spp", "demo", 256, 256, 1),
    ("demo_128x96_16spp", "demo", 128, 96, 16),
    ("demo_200x120_3spp", "demo", 200, 120, 3),        # ragged: not a multiple of any tile
    ("simple_96x96_4spp", "simple.scn", 96, 96, 4),
    ("cornell_96x96_4spp", "cornell.scn", 96, 96, 4),
    ("cornell_large_64x64_4spp", "cornell_large.scn", 64, 64, 4),
    ("caustic_96x64_8spp", "caustic.scn", 96, 64, 8),
    ("caustic3_64x64_8spp", "caustic3.scn", 64, 64, 8),
    ("demo_scn_64x64_4spp", "demo.scn", 64, 64, 4),
    ("complex_64x48_1spp", "complex.scn", 64, 48, 1),
    ("cornell_test_64x64_2spp", "cornell_test.scn", 64, 64, 2),
    ("complex_test_48x48_1spp", "complex_test.scn", 48, 48, 1),
    # the synthetic BASELINE scenes (SURVEY 8d 'Inputs'), as raytracing_simple_amd/scenes.py generates them and bench.py renders
    # them (the generator's arrays go to the kernel as they are: no loader, so no doubling): the reference's own outputs
    # for C3's scene, C5's closed mirror / glass box (paths of 8 bounces), the north-star 16-sphere scene, 256 scattered spheres
    ("c3_random_1024_64x40_2spp", "gen:random_spheres:1024", 64, 40, 2),
    ("c5_mirror_box_64_64x64_4spp", "gen:mirror_box:64", 64, 64, 4),
    ("c16_demo_plus_10_120x72_6spp", "gen:demo_plus:16", 120, 72, 6),
    ("random_256_96x64_4spp", "gen:random_spheres:256", 96, 64, 4),
]


def main():
    O.build_oracle(ref=True)
    ref = O.reference()
    for name, scene, w, h, spp in CASES:
        if scene == "demo":
            buf = np.zeros(64, O.SPHERE_DT)
            n = ref.ref_demo_scene(buf.ctypes.data_as(O.C.c_void_p), 64)
            sph, orig, target = buf[:n].copy(), O.DEMO_ORIG, O.DEMO_TARGET
        elif scene.startswith("gen:"):
            sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
            from raytracing_simple_amd import scenes           # numpy only: the generators never touch the library
            _, maker, count = scene.split(":")
            sph, orig, target = getattr(scenes, maker)(int(count))
            sph = np.ascontiguousarray(sph).view(O.SPHERE_DT) if sph.dtype != O.SPHERE_DT else sph
            orig, target = np.asarray(orig, np.float32), np.asarray(target, np.float32)
        else:
            sph, orig, target = O.ref_read_scene(os.path.join(SCN, scene))
        cam = np.zeros(15, np.float32)
        cam[0:3] = orig
        cam[3:6] = target
        ref.ref_camera_basis(cam.ctypes.data_as(O.C.c_void_p), w, h)
        out = O.ref_render(sph, cam, w, h, spp)
        new = dict(spheres=np.ascontiguousarray(sph).view(np.uint8), camera=cam, w=w, h=h, spp=spp,
                   pixels=out["pixels"], colors=out["colors"],
                   fnv_colors=O.fnv(out["colors"]), fnv_seeds=O.fnv(out["seeds"]),
                   fnv_pixels=O.fnv(out["pixels"]))
        path = os.path.join(OUT, name + ".npz")
        state = "written"
        if os.path.exists(path):        # a committed fixture the reference reproduces is left alone (zip members carry time stamps)
            old = np.load(path)
            same = set(old.files) == set(new) and all(np.array_equal(np.asarray(old[k]).view(np.uint8) if np.asarray(old[k]).dtype.kind == "f" else old[k],
                                                                     np.asarray(new[k]).view(np.uint8) if np.asarray(new[k]).dtype.kind == "f" else new[k]) for k in new)
            state = "unchanged" if same else "REWRITTEN (differs from the committed file)"
        if state != "unchanged":
            np.savez_compressed(path, **new)
        print(f"{name}: n={len(sph)} fnv(pixels)={O.fnv(out['pixels'])} {state}")
    # seed stream + camera pins
    sd = np.zeros(2 * 64 * 64, np.uint32)
    ref.ref_seeds_init(sd.ctypes.data_as(O.C.c_void_p), 64, 64)
    pins = {"seeds_first_8192": sd}
    for (w, h) in [(256, 256), (800, 600), (1920, 1080), (3840, 2160)]:
        s = np.zeros(2 * w * h, np.uint32)
        ref.ref_seeds_init(s.ctypes.data_as(O.C.c_void_p), w, h)
        pins[f"fnv_seeds_{w}x{h}"] = O.fnv(s)
        cam = np.zeros(15, np.float32)
        cam[0:3] = O.DEMO_ORIG
        cam[3:6] = O.DEMO_TARGET
        ref.ref_camera_basis(cam.ctypes.data_as(O.C.c_void_p), w, h)
        pins[f"camera_{w}x{h}"] = cam
    path = os.path.join(OUT, "host_pins.npz")
    old = np.load(path) if os.path.exists(path) else None
    if old is not None and set(old.files) == set(pins) and all(np.array_equal(np.asarray(old[k]).view(np.uint8) if np.asarray(old[k]).dtype.kind == "f" else old[k],
                                                                              np.asarray(pins[k]).view(np.uint8) if np.asarray(pins[k]).dtype.kind == "f" else pins[k]) for k in pins):
        print("host_pins unchanged")
    else:
        np.savez_compressed(path, **pins)
        print("host_pins written")


if __name__ == "__main__":
    main()
