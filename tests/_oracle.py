"""ctypes access to the CPU oracle (oracle/liboracle.so) and, where it has been built,
to the reference's own kernel compiled as host C++ (oracle/_ref/libref.so).

Test infrastructure only: nothing in raytracing_simple_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_ROOT = "/root/reference"

SPHERE_DT = np.dtype([("rad", "<f4"), ("p", "<f4", 3), ("e", "<f4", 3), ("c", "<f4", 3),
                      ("refl", "<i4")])
assert SPHERE_DT.itemsize == 44


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_calls", C.c_uint64),
                ("shadow_calls", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("rng_draws", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build_oracle(ref=True):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    if ref and os.path.isdir(REF_ROOT):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True)


_orc = None
_ref = None


def oracle():
    global _orc
    if _orc is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle(ref=False)
        if not _cpu_has_fma():
            raise RuntimeError("liboracle.so is built with -mfma; this CPU has no FMA")
        lib = C.CDLL(path)
        lib.orc_fnv1a64.restype = C.c_uint64
        lib.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.orc_get_random.restype = C.c_float
        lib.orc_sphere_intersect.restype = C.c_float
        lib.om_sinf.restype = C.c_float
        lib.om_sinf.argtypes = [C.c_float]
        lib.om_cosf.restype = C.c_float
        lib.om_cosf.argtypes = [C.c_float]
        lib.om_powf.restype = C.c_float
        lib.om_powf.argtypes = [C.c_float, C.c_float]
        lib.om_gammaf.restype = C.c_float
        lib.om_gammaf.argtypes = [C.c_float]
        lib.orc_to_int.argtypes = [C.c_float]
        lib.orc_math_mismatches.restype = C.c_uint64
        lib.orc_math_mismatches.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        lib.orc_glibc_rand_stream.argtypes = [C.c_void_p, C.c_size_t]
        _orc = lib
    return _orc


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref.so"))


def ref_tree_available():
    """The reference checkout itself (scene files, headers): only in the build container."""
    return ref_available() and os.path.isdir(REF_ROOT)


def reference():
    """The reference's kernel + host sources compiled in place (only in the build container)."""
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref.so"))
        lib.ref_get_random.restype = C.c_float
        lib.ref_sphere_intersect.restype = C.c_float
        _ref = lib
    return _ref


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % oracle().orc_fnv1a64(_ptr(a), a.nbytes)


# ---- inputs -----------------------------------------------------------------------------
DEMO_ORIG = (20.0, 100.0, 120.0)      # Main.cpp:83-84
DEMO_TARGET = (0.0, 25.0, 0.0)


def demo_spheres():
    """The built-in 6-sphere scene (values of Scene.cpp:5-12), as a structured array."""
    rows = [
        (1000, (0, -1000, 0), (0, 0, 0), (0.75, 0.75, 0.75), 0),
        (12, (40, 20, 0), (0, 0, 0), (0.9, 0, 0), 2),
        (11, (-35, 20, 0), (0, 0, 0), (0, 0.9, 0), 2),
        (10, (0, 25, -10), (0, 0, 0), (0, 0, 0.9), 2),
        (9, (20, 10, -5), (0, 0, 0), (0.9, 0, 0.9), 2),
        (7, (0, 60, 0), (12, 12, 12), (0, 0, 0), 0),
    ]
    return np.array(rows, dtype=SPHERE_DT)


def camera(orig, target, w, h):
    cam = np.zeros(15, np.float32)
    cam[0:3] = orig
    cam[3:6] = target
    oracle().orc_camera_basis(_ptr(cam), w, h)
    return cam


def seeds(w, h):
    s = np.zeros(2 * w * h, np.uint32)
    oracle().orc_seeds_init(_ptr(s), w, h)
    return s


def _as_spheres(a):
    """Sphere records as the 44-byte structured dtype (raw byte arrays, as the fixtures store them,
    are reinterpreted; anything else would silently change the sphere count)."""
    a = np.ascontiguousarray(a)
    if a.dtype != SPHERE_DT:
        if a.dtype != np.uint8 or a.size % 44:
            raise TypeError(f"spheres must be SPHERE_DT records or their raw bytes, not {a.dtype}[{a.size}]")
        a = a.reshape(-1).view(SPHERE_DT)
    return a


def render(spheres, cam, w, h, spp, first_sample=0, seeds_in=None, colors_in=None, threads=8,
           backend=0):
    """Oracle render: returns dict(pixels, colors, seeds, stats)."""
    lib = oracle()
    spheres = _as_spheres(spheres)
    sd = seeds(w, h) if seeds_in is None else seeds_in.copy()
    colors = np.zeros(3 * w * h, np.float32) if colors_in is None else colors_in.copy()
    pix = np.zeros(w * h, np.uint32)
    st = Stats()
    lib.orc_set_math_backend(backend)
    try:
        lib.orc_render(_ptr(colors), _ptr(sd), _ptr(spheres), C.c_uint32(len(spheres)), _ptr(cam),
                       w, h, first_sample, spp, _ptr(pix), threads, C.byref(st))
    finally:
        lib.orc_set_math_backend(0)
    return {"pixels": pix, "colors": colors, "seeds": sd, "stats": st.as_dict()}


def ref_render(spheres, cam, w, h, spp):
    """The reference kernel itself, spp launches in gid order (container only)."""
    lib = reference()
    spheres = _as_spheres(spheres)
    sd = np.zeros(2 * w * h, np.uint32)
    lib.ref_seeds_init(_ptr(sd), w, h)
    colors = np.zeros(3 * w * h, np.float32)
    pix = np.zeros(w * h, np.uint32)
    for s in range(spp):
        lib.ref_render_pass(_ptr(colors), _ptr(sd), _ptr(spheres), C.c_uint(len(spheres)),
                            _ptr(cam), w, h, s, _ptr(pix))
    return {"pixels": pix, "colors": colors, "seeds": sd}


def ref_render_mt(spheres, cam, w, h, spp, threads):
    """The reference kernel on `threads` host threads (same buffers as ref_render)."""
    lib = reference()
    spheres = _as_spheres(spheres)
    sd = np.zeros(2 * w * h, np.uint32)
    lib.ref_seeds_init(_ptr(sd), w, h)
    colors = np.zeros(3 * w * h, np.float32)
    pix = np.zeros(w * h, np.uint32)
    lib.ref_render_passes_mt(_ptr(colors), _ptr(sd), _ptr(spheres), C.c_uint(len(spheres)), _ptr(cam), w, h, 0, spp,
                             _ptr(pix), threads)
    return {"pixels": pix, "colors": colors, "seeds": sd}


def ref_read_scene(path, cap=8192):
    lib = reference()
    buf = np.zeros(cap, SPHERE_DT)
    o = np.zeros(3, np.float32)
    t = np.zeros(3, np.float32)
    n = lib.ref_read_scene(path.encode(), _ptr(buf), cap, _ptr(o), _ptr(t))
    if n < 0:
        raise ValueError("scene larger than cap")
    return buf[:n].copy(), tuple(float(v) for v in o), tuple(float(v) for v in t)


def psnr(a_pix, b_pix):
    """PSNR over the three 8-bit channels packed in the uint32 pixels."""
    a = np.ascontiguousarray(a_pix).view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    b = np.ascontiguousarray(b_pix).view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
