import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracing_simple_amd import api, host
W, H = 1920, 1080
n_ctx, interleave, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ctxs, pool = [], []
for _ in range(n_ctx):
    c = api.RtContext(W, H); c.set_scene(host.demo_scene()); c.set_camera(host.compute_camera(host.DEMO_ORIG, host.DEMO_TARGET, W, H)); ctxs.append(c)
    if interleave: pool.append(torch.cuda.Stream())
if not interleave: pool = [torch.cuda.Stream() for _ in range(n_ctx)]
own = [torch.cuda.ExternalStream(c.stream) for c in ctxs]
streams = pool if kind == "pool" else own
F = 2
def step(k):
    c, s = ctxs[k % F], streams[k % F]
    c.reset_async(s.cuda_stream); c.render_async(64, s.cuda_stream)
for k in range(4): step(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(24): step(k)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("n_ctx=%d interleave=%d streams=%s HWQ=%s: %.1f us/frame" % (n_ctx, interleave, kind, os.environ.get("GPU_MAX_HW_QUEUES", "default"), (t1 - t0) / 24 * 1e6), flush=True)
for c in ctxs: c.close()
