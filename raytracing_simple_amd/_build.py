"""Build recipe of the gfx950 libraries: explicit hipcc commands, in-tree output.

    python -m raytracing_simple_amd._build [--force]

  librt_hip.so       the product: exactly the entry points of include/rt_api.h (hidden visibility for
                     everything else) and only the kernel instances that ship
  librt_hip_diag.so  the same sources with -DRT_DIAGNOSTICS=1: every A/B / verification instance, the
                     exhaustive device-side checks, knobs and wall-clock logs of include/rt_debug.h
                     (tests/ and tools/ only)

The parity kernel translation unit and every host file are compiled with -ffp-contract=off;
only rt_kernel_fast.hip is allowed to fuse.  Device code is generated for gfx950 only."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librt_hip.so")
OUT_DIAG = os.path.join(HERE, "librt_hip_diag.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-fno-slp-vectorize",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
          "-fvisibility=hidden", "-DRT_BUILDING_LIBRARY=1",
          "-Wall", "-Wno-unused-function", "-Wno-sometimes-uninitialized", "-Wno-uninitialized"]
if os.environ.get("RT_BUILD_DEFINES"):        # experiments only (e.g. -DRT_BVH_LEAF=6): part of the build id like every flag
    COMMON = COMMON + os.environ["RT_BUILD_DEFINES"].split()
UNITS = [
    # source, extra flags
    ("rt_kernel_parity.hip", ["-ffp-contract=off"]),
    ("rt_kernel_fast.hip", ["-ffp-contract=fast"]),
    ("rt_api.hip", ["-ffp-contract=off"]),
    ("rt_launch.hip", ["-ffp-contract=off"]),
    ("rt_scene.hip", ["-ffp-contract=off"]),
    ("rt_debug.hip", ["-ffp-contract=off"]),
    ("rt_multi.hip", ["-ffp-contract=off"]),
    ("rt_bvh.hip", ["-ffp-contract=off"]),
    ("rt_host.cpp", ["-ffp-contract=off"]),
    ("rt_build_id.cpp", []),
]
DEPS = ["rt_device.h", "rt_internal.h", "rt_detmath.h", "rt_trace.inc.h", "rt_walk.inc.h", "rt_sched.inc.h", "rt_opts_reset.h",
        os.path.join("..", "..", "include", "rt_api.h"), os.path.join("..", "..", "include", "rt_debug.h")]


def source_hash():
    """What the libraries are built from, as one number: every source under csrc/, the two public headers and the compiler flags.
    It is compiled into both libraries (rt_build_id()), and tools/summarize_profile.py stamps every counter record with the id of
    the library it profiled -- so that bench.py can tell whether a committed instruction count still describes the code it runs."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h", ".cpp")))
    for path in [os.path.join(CSRC, n) for n in names] + [os.path.join(ROOT, "include", "rt_api.h"), os.path.join(ROOT, "include", "rt_debug.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    h.update(repr((COMMON, UNITS)).encode())
    return h.hexdigest()[:16]


def _build_id_header():
    """csrc/_obj/rt_build_id.h, rewritten only when the id changes (its timestamp is what makes rt_build_id.cpp stale)."""
    path = os.path.join(OBJ, "rt_build_id.h")
    text = '#define RT_BUILD_ID "%s"\n' % source_hash()
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


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


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def declared_symbols(*headers):
    """Every rt_* function an include/ header declares (comments stripped)."""
    import re
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def _version_script(path, headers):
    """Linker version script: the export table is exactly what the headers declare (kernel stubs, HIP's
    registration symbols and template instantiations of the C++ runtime stay local)."""
    text = "{\n  global:\n" + "".join("    %s;\n" % n for n in declared_symbols(*headers)) + "  local: *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def _flags_stamp(tag, defines):
    """csrc/_obj/flags<tag>.stamp, rewritten only when the compiler flags of this library change: every object depends on
    it, so a build with other flags (RT_BUILD_DEFINES=-DRT_BVH_LEAF=6 and back) recompiles every unit instead of relinking
    objects of the other configuration under the new build id."""
    path = os.path.join(OBJ, "flags" + tag + ".stamp")
    text = repr((COMMON, defines, UNITS))
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def _build_lib(cc, out, tag, defines, force, verbose, headers=("rt_api.h",)):
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.abspath(__file__), _flags_stamp(tag, defines)]
    id_header = _build_id_header()
    jobs, objs = [], []
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + tag + ".o")
        objs.append(op)
        if src == "rt_build_id.cpp":            # depends on nothing but the id itself
            if force or _stale(op, [sp, id_header]):
                jobs.append([cc] + COMMON + defines + ["-I" + OBJ, "-c", sp, "-o", op])
            continue
        if force or _stale(op, [sp] + deps):
            jobs.append([cc] + COMMON + defines + extra + ["-c", sp, "-o", op])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(lambda c: _run(c, verbose), jobs))
    vs = _version_script(os.path.join(OBJ, "exports" + tag + ".map"), headers)
    if force or _stale(out, objs + [vs]):
        _run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-Wl,--version-script=" + vs], verbose)
    return out


def kernel_metadata(lib=None):
    """What the gfx950 code objects inside a built library say about every kernel: registers, LDS, and the private
    segment (scratch) -- {symbol: {vgpr_count, sgpr_count, private_segment_fixed_size, vgpr_spill_count, ...}}.
    Read with the image's llvm-objdump / llvm-readelf in a scratch directory (tests/test_abi.py holds DESIGN.md's
    "no scratch, <= 96 registers" to it)."""
    import re
    import tempfile
    lib = lib or OUT
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(tmp, name)], check=True,
                                   capture_output=True, text=True).stdout
            for block in re.split(r"\n(?=\s+- \.agpr_count:)", notes)[1:]:      # (one block per kernel; the look-ahead keeps the entry's first key)
                fields = dict(re.findall(r"\.([a-z_]+):\s+(\S+)", block))
                out[fields["name"]] = {k: int(v) for k, v in fields.items() if re.fullmatch(r"\d+", v)}
    return out


def build(force=False, verbose=False, diag=True):
    """Compile whatever is out of date and link the libraries.  Returns the product library's path."""
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    _build_lib(cc, OUT, "", [], force, verbose)
    if diag:
        _build_lib(cc, OUT_DIAG, ".diag", ["-DRT_DIAGNOSTICS=1"], force, verbose, headers=("rt_api.h", "rt_debug.h"))
    build_harness(cc, force, verbose)
    return OUT


def build_harness(cc, force=False, verbose=False):
    """The native hosts on the C ABI: tools/rt_bench (the reference's Main.cpp without the window),
    tools/rt_inflight (bench.py's frames-in-flight loop), tools/rt_view (the display component, built
    only where freeglut's library exists -- its sources are syntax-checked by tests/test_adapter.py)."""
    inc = "-I" + os.path.join(ROOT, "include")
    link = ["-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN"]
    hdr = os.path.join(ROOT, "include", "rt_api.h")
    src = os.path.join(ROOT, "tools", "rt_bench.cpp")
    exe = os.path.join(HERE, "rt_bench")
    if force or _stale(exe, [src, OUT, hdr]):
        _run([cc, "-O2", "-std=c++17", inc, src, "-o", exe] + link, verbose)
    src2 = os.path.join(ROOT, "tools", "rt_inflight.cpp")
    exe2 = os.path.join(HERE, "rt_inflight")
    if force or _stale(exe2, [src2, OUT, hdr]):
        _run([cc, "--offload-arch=gfx950", "-O2", "-std=c++17", inc, src2, "-o", exe2] + link, verbose)
    src3 = os.path.join(ROOT, "tools", "view_headless.cpp")
    exe3 = os.path.join(HERE, "view_headless")
    deps3 = [src3, OUT, hdr, os.path.join(ROOT, "adapter", "ProgressiveRenderer.hpp"), os.path.join(ROOT, "adapter", "FrameExchange.hpp")]
    if force or _stale(exe3, deps3):
        _run([cc, "-O2", "-std=c++17", "-pthread", inc, src3, "-o", exe3] + link, verbose)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
