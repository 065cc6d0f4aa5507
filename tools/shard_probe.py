#!/usr/bin/env python3
"""Frames in flight on one rank's shard: python tools/shard_probe.py NRANKS F1,F2,...
Renders rank 0's interleaved share of the C2 frame with F contexts / streams in flight and prints
microseconds per frame (what one rank of an N-GPU job sustains before the gather)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracing_simple_amd import api, host
W, H = 1920, 1080
nranks = int(sys.argv[1])
for F in [int(v) for v in sys.argv[2].split(",")]:
    ctxs = []
    for _ in range(F):
        c = api.RtContext(W, H, rank=0, nranks=nranks, tile_rows=8)
        c.set_scene(host.demo_scene()); c.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W, H)); ctxs.append(c)
    streams = [torch.cuda.ExternalStream(c.stream) for c in ctxs]
    def step(k):
        c, s = ctxs[k % F], streams[k % F]
        c.reset_async(s.cuda_stream); c.render_async(64, s.cuda_stream)
    for k in range(2 * F): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 48
    for k in range(n): step(k)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("1/%d shard, %d frames in flight: %.1f us/frame" % (nranks, F, (t1 - t0) / n * 1e6), flush=True)
    for c in ctxs: c.close()
