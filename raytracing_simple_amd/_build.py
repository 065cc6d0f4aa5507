"""Build recipe of the gfx950 library (librt_hip.so): explicit hipcc commands, in-tree output.

    python -m raytracing_simple_amd._build

The parity kernel translation unit and every host file are compiled with -ffp-contract=off;
only rt_kernel_fast.hip is allowed to fuse.  Device code is generated for gfx950 only."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librt_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-fno-slp-vectorize",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
          "-Wall", "-Wno-unused-function", "-Wno-sometimes-uninitialized", "-Wno-uninitialized"]
UNITS = [
    # source, extra flags
    ("rt_kernel_parity.hip", ["-ffp-contract=off"]),
    ("rt_kernel_fast.hip", ["-ffp-contract=fast"]),
    ("rt_api.hip", ["-ffp-contract=off"]),
    ("rt_host.cpp", ["-ffp-contract=off"]),
]
DEPS = ["rt_device.h", "rt_detmath.h", "rt_trace.inc.h", "rt_sched.inc.h", "rt_opts_reset.h", os.path.join("..", "..", "include", "rt_api.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    """Compile whatever is out of date and link librt_hip.so.  Returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.abspath(__file__)]
    objs = []
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + deps):
            cmd = [cc] + COMMON + extra + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    if force or _stale(OUT, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    build_harness(cc, force, verbose)
    return OUT


def build_harness(cc, force=False, verbose=False):
    """tools/rt_bench: the headless C++ host (the reference's Main.cpp without the window)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "rt_bench.cpp")
    exe = os.path.join(HERE, "rt_bench")
    if force or _stale(exe, [src, OUT, os.path.join(root, "include", "rt_api.h")]):
        cmd = [cc, "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), src, "-o", exe,
               "-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # tools/rt_inflight: bench.py's frames-in-flight loop as a native host program (HIP only for the final wait)
    src2 = os.path.join(root, "tools", "rt_inflight.cpp")
    exe2 = os.path.join(HERE, "rt_inflight")
    if force or _stale(exe2, [src2, OUT, os.path.join(root, "include", "rt_api.h")]):
        cmd = [cc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), src2, "-o", exe2,
               "-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
