"""Host-side helpers either side of the render path, bound to the C ABI (rt_host.cpp):
camera basis, default seed stream, built-in scene, .scn reader; plus a PPM writer."""
import ctypes as C

import numpy as np

from .api import CAMERA_FLOATS, SPHERE_DT, RtError, _check, _ptr, load_library

DEMO_ORIG = (20.0, 100.0, 120.0)       # Main.cpp:83-84
DEMO_TARGET = (0.0, 25.0, 0.0)


def compute_camera(orig, target, w, h):
    """rt_compute_camera (computeCameraVariables, Utility.cpp:71-85) -> float32[15]."""
    cam = np.zeros(CAMERA_FLOATS, np.float32)
    cam[0:3] = orig
    cam[3:6] = target
    load_library().rt_compute_camera(_ptr(cam), w, h)
    return cam


def default_seeds(count):
    out = np.zeros(count, np.uint32)
    load_library().rt_default_seeds(_ptr(out), count)
    return out


def demo_scene():
    buf = np.zeros(6, SPHERE_DT)
    n = load_library().rt_demo_scene(_ptr(buf), 6)
    if n != 6:
        raise RtError(n, "rt_demo_scene")
    return buf


def read_scene(path, reference_doubling=True, cap=16384):
    """rt_read_scene -> (spheres, orig, target).  reference_doubling=True reproduces what the
    reference's loader hands to the kernel (N zeroed spheres in front of the N parsed ones)."""
    buf = np.zeros(cap, SPHERE_DT)
    n = C.c_uint32()
    o = np.zeros(3, np.float32)
    t = np.zeros(3, np.float32)
    _check(load_library().rt_read_scene(str(path).encode(), _ptr(buf), cap, C.byref(n), _ptr(o), _ptr(t),
                                        1 if reference_doubling else 0))
    return buf[:n.value].copy(), tuple(float(v) for v in o), tuple(float(v) for v in t)


def write_scene(path, spheres, orig, target):
    """Emit the reference's .scn text format (Utility.cpp:90-160)."""
    with open(path, "w") as f:
        f.write("camera %.9g %.9g %.9g  %.9g %.9g %.9g\n" % (*orig, *target))
        f.write("size %d\n" % len(spheres))
        for s in spheres:
            f.write("sphere %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %d\n" %
                    (s["rad"], *s["p"], *s["e"], *s["c"], int(s["refl"])))


def write_ppm(path, pixels, w, h):
    """Binary PPM of a packed pixel buffer.  Buffer row 0 is the BOTTOM of the image
    (the reference hands the buffer to glDrawPixels, SetupGL.cpp:59-63), so rows are flipped."""
    px = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(h, w)
    rgb = px.view(np.uint8).reshape(h, w, 4)[::-1, :, :3]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb).tobytes())


def write_png(path, pixels, w, h):
    """PNG of a packed pixel buffer (rows flipped like write_ppm).  Needs Pillow."""
    from PIL import Image
    px = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(h, w)
    rgb = np.ascontiguousarray(px.view(np.uint8).reshape(h, w, 4)[::-1, :, :3])
    Image.fromarray(rgb, "RGB").save(path)


def psnr(a_pix, b_pix):
    a = np.ascontiguousarray(a_pix, dtype=np.uint32).view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    b = np.ascontiguousarray(b_pix, dtype=np.uint32).view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0.0 else 10.0 * np.log10(255.0 ** 2 / mse)
