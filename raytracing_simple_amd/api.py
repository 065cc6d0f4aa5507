"""ctypes binding of include/rt_api.h (one Python method per C entry point)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SPHERE_DT = np.dtype([("rad", "<f4"), ("p", "<f4", 3), ("e", "<f4", 3), ("c", "<f4", 3),
                      ("refl", "<i4")])           # include/rt_api.h rt_sphere, 44 bytes
CAMERA_FLOATS = 15                                # rt_camera: orig,target,dir,x,y

RT_MODE_PARITY, RT_MODE_FAST = 0, 1
DIFF, SPEC, REFR = 0, 1, 2

# every symbol include/rt_api.h declares = the export table of librt_hip.so (tests/test_abi.py)
SYMBOLS = ["rt_render", "rt_release_cache", "rt_create", "rt_create_multi", "rt_create_multi_on", "rt_shard_count", "rt_last_kernel", "rt_scene_choice",
           "rt_create_sharded", "rt_destroy", "rt_set_scene", "rt_update_spheres_async",
           "rt_set_camera", "rt_set_mode", "rt_reset", "rt_reset_async", "rt_render_pass", "rt_render_async",
           "rt_pin_output", "rt_set_pixel_write", "rt_read_pixels", "rt_read_pixels_async", "rt_throttle", "rt_device_pixels", "rt_set_pixel_buffer", "rt_stream",
           "rt_local_rows", "rt_current_sample", "rt_read_colors",
           "rt_read_seeds", "rt_get_stats", "rt_last_error", "rt_deinterleave_rows", "rt_compute_camera",
           "rt_default_seeds", "rt_demo_scene", "rt_read_scene", "rt_build_id"]
# include/rt_debug.h: what librt_hip_diag.so exports on top of that
DEBUG_SYMBOLS = ["rt_debug_variant_count", "rt_debug_instance", "rt_debug_instance_name", "rt_debug_shard_kernel", "rt_debug_break_gather", "rt_debug_set_rccl_library", "rt_debug_stage_tables", "rt_debug_eval", "rt_debug_sqrt_mismatches", "rt_debug_hitpost_mismatches",
                 "rt_debug_rcp_probe", "rt_debug_set_regen_gate", "rt_debug_set_mat_lds_limit", "rt_debug_set_persist",
                 "rt_debug_set_ncus", "rt_debug_set_coop_min", "rt_debug_set_bvh", "rt_debug_set_tree_shape", "rt_debug_set_walk", "rt_debug_set_walk_round", "rt_debug_bvh_pick", "rt_debug_tree_estimate", "rt_debug_set_choice_estimate", "rt_debug_create_breakdown", "rt_debug_walk_rays", "rt_debug_read_bvh", "rt_debug_read_packed_pairs", "rt_debug_set_tile_order", "rt_debug_read_tile_order", "rt_debug_set_wg_waves", "rt_debug_counters", "rt_debug_counters_raw",
                 "rt_debug_reset_by_copy", "rt_debug_probe_seeds", "rt_debug_sidelog_read", "rt_debug_timelog_enable", "rt_debug_timelog_tag",
                 "rt_debug_timelog_read", "rt_debug_wavelog_read"]


class RtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"rt status {code}: {text}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("rng_draws", C.c_uint64), ("launches", C.c_uint64),
                ("last_kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _Scene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("count", C.c_uint32)]


def lib_path(diag=False):
    return os.path.join(_HERE, "librt_hip_diag.so" if diag else "librt_hip.so")


_libs = {}


def load_library(diag=False):
    """Load librt_hip.so (built in-tree by raytracing_simple_amd._build).  Raises if absent:
    the HIP library IS the product, there is nothing to fall back to.  diag=True loads the
    diagnostics build instead (librt_hip_diag.so: include/rt_debug.h on top of the same ABI; tests and
    tools only -- bench.py and the product paths never do)."""
    if diag in _libs:
        return _libs[diag]
    path = lib_path(diag)
    if not os.path.exists(path):
        raise RtError(-2, f"{path} is not built; run `python -m raytracing_simple_amd._build` "
                          "(needs hipcc) -- there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    u64p = C.POINTER(C.c_ulonglong)
    sig = {
        "rt_render": (i32, [vp, vp, vp, i32, i32, i32]),
        "rt_release_cache": (None, []),
        "rt_create_multi": (i32, [C.POINTER(vp), i32, i32, i32]),
        "rt_create_multi_on": (i32, [C.POINTER(vp), i32, i32, C.POINTER(i32), i32, i32]),
        "rt_shard_count": (i32, [vp]),
        "rt_last_kernel": (C.c_char_p, [vp]),
        "rt_scene_choice": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rt_build_id": (C.c_char_p, []),
        "rt_update_spheres_async": (i32, [vp, u32, u32, vp, vp]),
        "rt_read_pixels": (i32, [vp, vp]),
        "rt_read_pixels_async": (i32, [vp, vp, vp]),
        "rt_throttle": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "rt_deinterleave_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        "rt_create": (i32, [C.POINTER(vp), i32, i32]),
        "rt_create_sharded": (i32, [C.POINTER(vp), i32, i32, i32, i32, i32, i32]),
        "rt_destroy": (None, [vp]),
        "rt_set_scene": (i32, [vp, vp, u32]),
        "rt_set_camera": (i32, [vp, vp]),
        "rt_set_mode": (i32, [vp, i32]),
        "rt_reset": (i32, [vp]),
        "rt_reset_async": (i32, [vp, vp]),
        "rt_render_pass": (i32, [vp, vp, i32]),
        "rt_render_async": (i32, [vp, i32, vp]),
        "rt_device_pixels": (i32, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "rt_set_pixel_buffer": (i32, [vp, vp, sz]),
        "rt_pin_output": (i32, [vp, vp, sz]),
        "rt_set_pixel_write": (i32, [vp, i32]),
        "rt_stream": (vp, [vp]),
        "rt_local_rows": (i32, [vp]),
        "rt_current_sample": (i32, [vp]),
        "rt_read_colors": (i32, [vp, vp]),
        "rt_read_seeds": (i32, [vp, vp]),
        "rt_get_stats": (i32, [vp, C.POINTER(Stats)]),
        "rt_last_error": (C.c_char_p, []),
        "rt_compute_camera": (None, [vp, i32, i32]),
        "rt_default_seeds": (None, [vp, sz]),
        "rt_demo_scene": (i32, [vp, u32]),
        "rt_read_scene": (i32, [C.c_char_p, vp, u32, C.POINTER(u32), vp, vp, i32]),
    }
    if diag:
        sig.update({
            "rt_debug_variant_count": (i32, [i32]),
            "rt_debug_instance": (i32, [C.c_char_p]),
            "rt_debug_instance_name": (C.c_char_p, [i32, i32]),
            "rt_debug_shard_kernel": (C.c_char_p, [vp, i32]),
            "rt_debug_break_gather": (i32, [vp]),
            "rt_debug_set_rccl_library": (i32, [C.c_char_p, i32]),
            "rt_debug_stage_tables": (i32, [vp, i32, i32]),
            "rt_debug_eval": (i32, [i32, vp, vp, sz]),
            "rt_debug_sqrt_mismatches": (C.c_longlong, []),
            "rt_debug_hitpost_mismatches": (C.c_longlong, []),
            "rt_debug_rcp_probe": (i32, [vp]),
            "rt_debug_set_regen_gate": (i32, [vp, i32]),
            "rt_debug_set_mat_lds_limit": (i32, [vp, i32]),
            "rt_debug_set_persist": (i32, [vp, i32]),
            "rt_debug_set_ncus": (i32, [vp, i32]),
            "rt_debug_set_coop_min": (i32, [vp, i32]),
            "rt_debug_set_bvh": (i32, [vp, i32, i32]),
            "rt_debug_set_tree_shape": (i32, [vp, i32]),
            "rt_debug_set_walk": (i32, [vp, i32, i32, i32]),
            "rt_debug_bvh_pick": (i32, [vp]),
            "rt_debug_tree_estimate": (i32, [vp, C.POINTER(C.c_double)]),
            "rt_debug_set_choice_estimate": (i32, [vp, i32]),
            "rt_debug_create_breakdown": (i32, [C.POINTER(C.c_double)]),
            "rt_debug_walk_rays": (i32, [vp, vp, u32, vp]),
            "rt_debug_set_walk_round": (i32, [vp, i32]),
            "rt_debug_read_bvh": (i32, [vp, vp, u32, vp]),
            "rt_debug_read_packed_pairs": (i32, [vp, vp, u32, C.POINTER(u32)]),
            "rt_debug_set_tile_order": (i32, [vp, i32]),
            "rt_debug_set_wg_waves": (i32, [vp, i32]),
            "rt_debug_read_tile_order": (i32, [vp, vp, vp, u32, C.POINTER(u32), C.POINTER(i32)]),
            "rt_debug_counters": (i32, [vp, vp]),
            "rt_debug_counters_raw": (i32, [vp, vp]),
            "rt_debug_reset_by_copy": (i32, [vp, vp, i32]),
            "rt_debug_probe_seeds": (i32, [vp, vp, i32]),
            "rt_debug_sidelog_read": (i32, [vp, u32, vp, vp]),
            "rt_debug_timelog_enable": (i32, [vp, u32, u32]),
            "rt_debug_timelog_tag": (i32, [vp, C.c_ulonglong]),
            "rt_debug_timelog_read": (i32, [vp, vp, u32, C.POINTER(u32)]),
            "rt_debug_wavelog_read": (i32, [vp, vp, u32]),
        })
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _libs[diag] = lib
    return lib


def _check(rc, lib=None):
    if rc != 0:
        raise RtError(rc, (lib or load_library()).rt_last_error().decode(errors="replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_spheres(spheres):
    """Sphere records as the 44-byte structured dtype; raw bytes (uint8, a multiple of 44) are
    reinterpreted, anything else is refused rather than silently given another sphere count."""
    a = np.ascontiguousarray(spheres)
    if a.dtype != SPHERE_DT:
        if a.dtype != np.uint8 or a.size % SPHERE_DT.itemsize:
            raise TypeError(f"spheres must be SPHERE_DT records or their raw bytes, not {a.dtype}[{a.size}]")
        a = a.reshape(-1).view(SPHERE_DT)
    return a


def as_camera(cam):
    a = np.ascontiguousarray(cam, dtype=np.float32).reshape(-1)
    if a.size != CAMERA_FLOATS:
        raise ValueError("camera = 15 floats: orig, target, dir, x, y")
    return a


def render(spheres, cam, w, h, spp):
    """rt_render: the one-shot headline call.  Returns uint32[h*w] (row 0 = bottom)."""
    lib = load_library()
    sph = as_spheres(spheres)
    cam = as_camera(cam)
    out = np.zeros(w * h, np.uint32)
    scene = _Scene(sph.ctypes.data, len(sph))
    _check(lib.rt_render(C.addressof(scene), _ptr(cam), _ptr(out), w, h, spp))
    return out


class RtContext:
    """One rt_ctx (= one OpenCLConfigBuffer of the reference).

    devices=[...] makes it a multi-device context (rt_create_multi_on): the image sharded over those
    HIP devices of this process, one RCCL gather per frame (a device listed twice = the one-GPU
    rehearsal of that path).  diag=True binds the diagnostics library (tests / tools only)."""

    def __init__(self, w, h, device=0, rank=0, nranks=1, tile_rows=8, devices=None, diag=False):
        self._lib = load_library(diag)
        self._h = C.c_void_p()
        self.w, self.h = w, h
        self.rank, self.nranks, self.tile_rows = rank, nranks, tile_rows
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            self._check(self._lib.rt_create_multi_on(C.byref(self._h), w, h, arr, len(devices), tile_rows))
            self.rank, self.nranks = 0, 1           # the front of a multi-device context is the whole image
        else:
            self._check(self._lib.rt_create_sharded(C.byref(self._h), w, h, device, rank, nranks, tile_rows))

    def _check(self, rc):
        _check(rc, self._lib)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: modules may already be gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- state ---------------------------------------------------------------------------
    def set_scene(self, spheres):
        sph = as_spheres(spheres)
        self._check(self._lib.rt_set_scene(self._h, _ptr(sph), len(sph)))

    def set_camera(self, cam):
        cam = as_camera(cam)
        self._check(self._lib.rt_set_camera(self._h, _ptr(cam)))

    def set_mode(self, mode):
        self._check(self._lib.rt_set_mode(self._h, mode))

    def reset(self):
        self._check(self._lib.rt_reset(self._h))

    @property
    def local_rows(self):
        return self._lib.rt_local_rows(self._h)

    @property
    def current_sample(self):
        return self._lib.rt_current_sample(self._h)

    # --- rendering ----------------------------------------------------------------------
    def render_pass(self, n_samples, copy=True, out=None):
        """n_samples reference passes in one launch; returns the local pixel rows (uint32),
        written into `out` when one is given (a C-contiguous uint32 array of that size)."""
        if out is None:
            out = np.zeros(self.local_rows * self.w, np.uint32) if copy else None
        elif out.dtype != np.uint32 or out.size < self.local_rows * self.w or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint32 array of local_rows * w elements")
        self._check(self._lib.rt_render_pass(self._h, _ptr(out) if out is not None else None, n_samples))
        return out

    def set_pixel_write(self, enable):
        self._check(self._lib.rt_set_pixel_write(self._h, 1 if enable else 0))

    def pin_output(self, out):
        """Page-lock `out` (the array later passed to render_pass(out=...)) for full-rate readback;
        None unpins.  The array must outlive the pin."""
        if out is None:
            self._check(self._lib.rt_pin_output(self._h, None, 0))
        else:
            self._check(self._lib.rt_pin_output(self._h, _ptr(out), out.size))

    def read_pixels(self):
        """rt_read_pixels: the up-to-date packed frame (packs it from the running average first if the
        last launches ran with the pixel store off)."""
        out = np.zeros(self.local_rows * self.w, np.uint32)
        self._check(self._lib.rt_read_pixels(self._h, _ptr(out)))
        return out

    def read_pixels_async(self, out, stream=None):
        """rt_read_pixels_async: queue the copy of the up-to-date frame into `out` (page-lock it with pin_output)."""
        self._check(self._lib.rt_read_pixels_async(self._h, _ptr(out), C.c_void_p(stream or 0)))

    def throttle(self, max_in_flight):
        """rt_throttle: wait until at most `max_in_flight` render_async launches are unfinished; returns the device
        time per pass (ms) of the most recent finished one (0.0 if none)."""
        ms = C.c_double()
        self._check(self._lib.rt_throttle(self._h, max_in_flight, C.byref(ms)))
        return ms.value

    def update_spheres(self, first, spheres, stream=None):
        """rt_update_spheres_async: replace spheres [first, first+len) of the current scene."""
        sph = as_spheres(spheres)
        self._check(self._lib.rt_update_spheres_async(self._h, first, len(sph), _ptr(sph), C.c_void_p(stream or 0)))

    @property
    def last_kernel(self):
        """Symbol of the kernel instance the last launch used (rt_last_kernel)."""
        return self._lib.rt_last_kernel(self._h).decode()

    def scene_choice(self):
        """rt_scene_choice: {'picked': 0 | 'hierarchy' | 'sweep', and the two measured ms per pass}."""
        a, b = C.c_double(), C.c_double()
        k = self._lib.rt_scene_choice(self._h, C.byref(a), C.byref(b))
        return {"picked": {0: None, 1: "hierarchy", 2: "sweep"}.get(k), "hierarchy_ms_per_pass": a.value, "sweep_ms_per_pass": b.value}

    @property
    def shard_count(self):
        return self._lib.rt_shard_count(self._h)

    def reset_async(self, stream=None):
        self._check(self._lib.rt_reset_async(self._h, C.c_void_p(stream or 0)))

    def render_async(self, n_samples, stream=None):
        self._check(self._lib.rt_render_async(self._h, n_samples, C.c_void_p(stream or 0)))

    def set_pixel_buffer(self, dptr, count):
        """Later launches write their packed pixels to this device address (None = own buffer)."""
        self._check(self._lib.rt_set_pixel_buffer(self._h, C.c_void_p(dptr or 0), count))

    @property
    def stream(self):
        """Raw hipStream_t of the context's own stream (wrap with torch.cuda.ExternalStream)."""
        return self._lib.rt_stream(self._h)

    def device_pixels(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.rt_device_pixels(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def device_pixels_array(self):
        """The local pixel rows as a zero-copy __cuda_array_interface__ object (int32 [rows, w]),
        for torch.as_tensor(..., device='cuda') in the multi-GPU gather."""
        ptr, n = self.device_pixels()

        class _View:
            pass

        v = _View()
        v.__cuda_array_interface__ = {"shape": (n // self.w, self.w), "typestr": "<i4",
                                      "data": (ptr, False), "version": 2, "strides": None}
        v._owner = self
        return v

    def read_colors(self):
        out = np.zeros(3 * self.w * self.h, np.float32)
        self._check(self._lib.rt_read_colors(self._h, _ptr(out)))
        return out

    def read_seeds(self):
        out = np.zeros(2 * self.w * self.h, np.uint32)
        self._check(self._lib.rt_read_seeds(self._h, _ptr(out)))
        return out

    def stats(self):
        st = Stats()
        self._check(self._lib.rt_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    # --- sharding helpers ---------------------------------------------------------------
    def local_row_map(self):
        """global row index of every local row, in local order."""
        return local_rows_of(self.h, self.rank, self.nranks, self.tile_rows)


def local_rows_of(h, rank, nranks, tile_rows):
    rows = []
    n_tiles = (h + tile_rows - 1) // tile_rows
    for t in range(rank, n_tiles, nranks):
        rows.extend(range(t * tile_rows, min(h, (t + 1) * tile_rows)))
    return np.asarray(rows, dtype=np.int64)


def deinterleave_rows(full_ptr, gathered_ptr, w, h, nranks, tile_rows, pad_rows, device=0, stream=None):
    """rt_deinterleave_rows on raw device pointers (the gather root's frame assembly)."""
    _check(load_library().rt_deinterleave_rows(C.c_void_p(full_ptr), C.c_void_p(gathered_ptr), w, h, nranks, tile_rows,
                                               pad_rows, device, C.c_void_p(stream or 0)))


def build_id(diag=False):
    """rt_build_id(): the identity of the sources and flags the loaded library was built from."""
    return load_library(diag).rt_build_id().decode()


def instance_mode(kernel_symbol):
    """The rt_set_mode value of the diagnostics library that selects the instance with this kernel symbol
    (rt_debug_instance: rows of the instance tables are found by name, never by number)."""
    lib = load_library(diag=True)
    mode = lib.rt_debug_instance(kernel_symbol.encode())
    if mode < 0:
        raise RtError(mode, lib.rt_last_error().decode(errors="replace"))
    return mode


def instance_names(fast=False):
    """Kernel symbols of every instance of the diagnostics library's parity (or fast) table, in row order."""
    lib = load_library(diag=True)
    return [lib.rt_debug_instance_name(1 if fast else 0, k).decode() for k in range(lib.rt_debug_variant_count(1 if fast else 0))]


def debug_eval(op, values):
    lib = load_library(diag=True)
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.zeros_like(v)
    _check(lib.rt_debug_eval(op, _ptr(v), _ptr(out), v.size), lib)
    return out
