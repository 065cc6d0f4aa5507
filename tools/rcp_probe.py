"""Exhaustive probe of lean reciprocal candidates against the correctly rounded 1.f/x (GPU)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_simple_amd import api
lib = api.load_library(diag=True)
buf = (C.c_ulonglong * 1024)()
rc = lib.rt_debug_rcp_probe(buf)
assert rc == 0, rc
names = ["v_rcp", "1 step", "2 steps", "2 steps (e1*r0)"]
for v in range(4):
    h = [buf[v * 256 + e] for e in range(256)]
    bad = [(e, c) for e, c in enumerate(h) if c]
    print(names[v], "total", sum(h), "clean exponents:", [e for e in range(256) if not h[e]][:3], "...",
          "first/last clean run:", end=" ")
    clean = [e for e in range(256) if not h[e]]
    print((min(clean), max(clean)) if clean else None, "dirty inside:", [(e, c) for e, c in bad if clean and min(clean) < e < max(clean)][:10])
