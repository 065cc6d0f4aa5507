#!/usr/bin/env python3
"""`python tools/isa_compare.py A.so B.so`: every gfx950 kernel of two builds of the library, disassembled (llvm-objdump of the code objects inside),
addresses and encodings stripped, compared instruction for instruction -- which kernels a change of the sources touched."""
import subprocess, sys, os, re, tempfile, shutil, hashlib
def kernels(lib):
    tmp = tempfile.mkdtemp()
    shutil.copy(lib, tmp)
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", os.path.basename(lib)], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f: continue
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
        cur = None
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m: cur = m.group(1); out[cur] = []; continue
            if cur and "//" in line:
                ins = line.split("//")[0].strip()
                ins = re.sub(r"<[^>]+>", "", ins)
                out[cur].append(ins)
    shutil.rmtree(tmp)
    return out
a = kernels(sys.argv[1]); b = kernels(sys.argv[2])
for k in sorted(set(a) | set(b)):
    ha = hashlib.md5("\n".join(a.get(k, [])).encode()).hexdigest()[:8] if k in a else None
    hb = hashlib.md5("\n".join(b.get(k, [])).encode()).hexdigest()[:8] if k in b else None
    print("%-60s %6s %6s %s" % (k[:60], len(a.get(k, [])), len(b.get(k, [])), "same" if ha == hb else "DIFFERENT"))
