"""What one rank of an N-GPU run sustains: shard = 1/N of the frame, F frames in flight on F streams.
usage: enqueue_cost.py [N] [F]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracing_simple_amd import api, host
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for F in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 2, 3, 4]):
    ctxs, bufs, streams = [], [], []
    for _ in range(F):
        c = api.RtContext(W, H, rank=0, nranks=N, tile_rows=8)
        c.set_scene(host.demo_scene()); c.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W, H))
        ctxs.append(c); bufs.append(torch.zeros((c.local_rows, W), dtype=torch.int32, device="cuda")); streams.append(torch.cuda.Stream())
    def step(k):
        c, b, s = ctxs[k % F], bufs[k % F], streams[k % F]
        c.set_pixel_buffer(b.data_ptr(), b.numel()); c.reset_async(s.cuda_stream); c.render_async(64, s.cuda_stream)
    for k in range(2 * F): step(k)
    torch.cuda.synchronize()
    K = 120
    t0 = time.perf_counter()
    for k in range(K): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("N=%d shard, %d frames in flight: enqueue %.1f us/step, GPU %.1f us/frame (ideal %.1f)" %
          (N, F, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, 3660.0 / N), flush=True)
    for c in ctxs: c.close()
