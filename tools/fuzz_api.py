#!/usr/bin/env python3
"""Random sequences of context operations against the oracle: passes of random length (fused
launches continuing the running average), resets, pixel-write off/on, pinned and unpinned output
buffers, caller-owned device buffers, mode switches back to parity, and sharded contexts whose rows
are reassembled; round 2: rt_read_pixels after passes without pixel stores, rt_update_spheres_async (same
records: upload + device-side table build in mid-sequence), launches and resets that hop between the null
stream, the context's stream and a foreign stream (the library chains them), and multi-device contexts
(one-GPU rehearsal: shards on device 0).  After every sequence pixels, colours and seeds must equal the oracle's for the
same total number of passes.    python tools/fuzz_api.py FIRST COUNT"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _oracle as O
from raytracing_simple_amd import api, host, scenes
from raytracing_simple_amd import dist as rdist

SCENES = [lambda: (host.demo_scene(), host.DEMO_ORIG, host.DEMO_TARGET), lambda: scenes.demo_plus(16),
          lambda: scenes.random_spheres(40), lambda: scenes.mirror_box(20),
          # large enough for the hierarchy: the first two launches of the sequence are its timing probes
          lambda: scenes.random_spheres(150), lambda: scenes.mirror_box(90)]
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    sph, orig, target = SCENES[seed % 6]()
    w, h = int(rng.integers(9, 90)), int(rng.integers(5, 70))
    cam = host.compute_camera(orig, target, w, h)
    nranks = int(rng.choice([1, 1, 2, 3]))
    multi = nranks == 1 and seed % 5 == 0
    if multi:
        ctxs = [api.RtContext(w, h, devices=[0] * int(rng.integers(1, 5)), tile_rows=8)]
    else:
        ctxs = [api.RtContext(w, h, rank=r, nranks=nranks, tile_rows=8) for r in range(nranks)]
    side = torch.cuda.Stream()
    for c in ctxs:
        c.set_scene(sph); c.set_camera(cam)
    total = 0
    outs = [np.zeros(c.local_rows * w, np.uint32) for c in ctxs]
    ext = [torch.zeros(max(c.local_rows * w, 1), dtype=torch.int32, device="cuda") for c in ctxs]
    log = []
    for _ in range(int(rng.integers(2, 9))):
        op = int(rng.integers(0, 12))
        if multi and op in (3, 11):
            op = 7                                      # no caller-owned pixel buffer / foreign stream on a multi-device context
        n = int(rng.integers(0, 6))
        log.append((op, n))
        for k, c in enumerate(ctxs):
            if op == 0:
                if n % 2:
                    c.reset()
                else:                                   # the device-side reset (only this rank's row tiles)
                    c.reset_async(c.stream); torch.cuda.synchronize()
            elif op == 1:
                c.set_pixel_write(False); c.render_pass(n, copy=False); c.set_pixel_write(True)
            elif op == 2:
                c.pin_output(outs[k]); c.render_pass(n, out=outs[k]); c.pin_output(None)
            elif op == 3:
                c.set_pixel_buffer(ext[k].data_ptr(), ext[k].numel()); c.render_pass(n, copy=False); c.set_pixel_buffer(0, 0)
            elif op == 4:
                c.set_mode(api.RT_MODE_FAST); c.set_mode(api.RT_MODE_PARITY); c.render_pass(n, out=outs[k])
            elif op == 5:
                c.render_async(n, c.stream); torch.cuda.synchronize()
            elif op == 8:
                c.set_pixel_write(False); c.render_async(n, c.stream); c.read_pixels(); c.set_pixel_write(True)
            elif op == 9:
                k0 = int(rng.integers(0, len(sph))); c.update_spheres(k0, sph[k0:k0 + 1 + n], None if multi else c.stream); c.render_pass(n, copy=False)
            elif op == 10:
                c.render_async(n, None); c.render_async(1, None if multi else c.stream); torch.cuda.synchronize()   # null stream, then own
            elif op == 11:
                c.render_async(n, side.cuda_stream); c.render_pass(0); c.render_async(0, c.stream)
            else:
                c.render_pass(n, out=outs[k])
        total = 0 if op == 0 else total + n + (1 if op == 10 else 0)
    # one more plain pass so that the pixel buffer reflects the running average of all passes
    parts = [c.render_pass(1) for c in ctxs]
    total += 1
    px = rdist.assemble_numpy(parts, h, w, nranks, 8)
    want = O.render(sph, cam, w, h, total)
    cols = ctxs[0].read_colors(); seeds = ctxs[0].read_seeds()
    ok = np.array_equal(px, want["pixels"])
    if nranks == 1:
        ok = ok and np.array_equal(cols.view(np.uint32), want["colors"].view(np.uint32)) and np.array_equal(seeds, want["seeds"])
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, (w, h), "ranks", nranks, "ops", log, "total passes", total, flush=True)
    for c in ctxs:
        c.close()
print("api fuzz seeds", first, "..", first + count - 1, "mismatches:", bad)
