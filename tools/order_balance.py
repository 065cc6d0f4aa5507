#!/usr/bin/env python3
"""Per-XCD balance of the heavy-first tile order (rt_order_tiles_kernel): the summed tile costs of the workgroup numbers equal
modulo 8, for the plain order (homes = 1) and for regions kept on one XCD (homes = 8): python tools/order_balance.py [c2|c16|c3]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
maker, w, h, spp = CONFIGS[name]
sph, orig, target = maker()
cam = host.compute_camera(orig, target, w, h)
lib = api.load_library(diag=True)
for homes in (1, 8):
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(lib.rt_debug_set_tile_order(ctx._h, 1 | (homes << 8)))
        ctx.set_scene(sph); ctx.set_camera(cam)
        for _ in range(9):
            ctx.reset(); ctx.render_pass(spp, copy=False)
        cap = 1 << 20
        order = np.zeros(cap, np.uint32); cost = np.zeros(cap, np.uint32)
        n = C.c_uint32(0); valid = C.c_int(0)
        ctx._check(lib.rt_debug_read_tile_order(ctx._h, order.ctypes.data_as(C.POINTER(C.c_uint32)), cost.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(n), C.byref(valid)))
        n = n.value
        order, cost = order[:n], cost[:n].astype(np.float64)
        if not valid.value:
            print(json.dumps({'config': name, 'homes': homes, 'tiles': n, 'order': 'none in use'})); continue
        assert sorted(order.tolist()) == list(range(n)), 'not a permutation'
        per = np.array([cost[order[x::8]].sum() for x in range(8)])
        print(json.dumps({"config": name, "homes": homes, "tiles": n, "ms": round(ctx.stats()["last_kernel_ms"], 3),
                          "per_xcd_cost_over_mean": [round(v, 4) for v in (per / per.mean()).tolist()]}), flush=True)
