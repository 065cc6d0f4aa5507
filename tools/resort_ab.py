import sys, os, statistics, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracing_simple_amd import api, host
from tools.ab_bench import CONFIGS
import numpy as np
for cname in sys.argv[1].split(","):
    maker, w, h, spp = CONFIGS[cname]
    sph, orig, target = maker()
    cam = host.compute_camera(orig, target, w, h)
    res = {}
    ctxs = {}
    for arm in ("priced_order_kept", "resorted_from_full_frame"):
        c = api.RtContext(w, h); c.set_scene(sph); c.set_camera(cam)
        c.render_pass(spp, copy=False)          # first frame: pricing + sort
        if arm == "resorted_from_full_frame":
            c.set_camera(host.compute_camera((orig[0] + 0.001, orig[1], orig[2]), target, w, h)); c.set_camera(cam)   # marks the order stale: next long launch sorts again
        ctxs[arm] = c
        res[arm] = []
    for r in range(14):
        for arm, c in ctxs.items():
            c.reset(); c.render_pass(spp, copy=False)
            if r >= 6: res[arm].append(c.stats()["last_kernel_ms"])
    # unseen passes
    un = {}
    for arm, c in ctxs.items():
        c.reset(); c.render_pass(spp, copy=False)
        t = []
        for _ in range(6):
            c.render_pass(spp, copy=False); t.append(c.stats()["last_kernel_ms"])
        un[arm] = statistics.median(t)
    print(json.dumps({"config": cname, **{a: round(statistics.median(v), 4) for a, v in res.items()}, "unseen": {a: round(v, 4) for a, v in un.items()}}), flush=True)
