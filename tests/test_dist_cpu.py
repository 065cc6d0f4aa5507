"""The N>1 path on CPU: two processes, gloo, interleaved row-tile shards -> one gather ->
de-interleave.  The local buffers are filled from a known image instead of a render, so this
covers exactly the distributed part (partition arithmetic + collective + reassembly)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_simple_amd import api
from raytracing_simple_amd import dist as rdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, tile_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = (np.arange(h * w, dtype=np.int64) * 2654435761 % (1 << 31)).astype(np.int32).reshape(h, w)
        g = rdist.FrameGatherer(h, w, rank, world, tile_rows, "cpu")
        rows = api.local_rows_of(h, rank, world, tile_rows)
        assert g.n_local == len(rows)
        g.local.zero_()
        g.local[: len(rows)] = torch.from_numpy(full[rows])
        out = g.gather()
        dist.barrier()
        ok = True
        if rank == 0:
            ok = bool(np.array_equal(out.numpy(), full))
        else:
            assert out is None
        # two slots, asynchronous: frame 1 is queued before frame 0 is waited for
        g2 = rdist.FrameGatherer(h, w, rank, world, tile_rows, "cpu", slots=2)
        for k in (0, 1):
            g2.local_slot(k).zero_()
            g2.local_slot(k)[: len(rows)] = torch.from_numpy(full[rows] + k)
            assert g2.gather(k, async_op=True) is None
        for k in (0, 1):
            out = g2.wait(k)
            if rank == 0:
                ok = ok and bool(np.array_equal(out.numpy(), full + k))
        dist.barrier()
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,tile_rows", [(64, 40, 8), (50, 33, 8), (17, 5, 16), (8, 8, 8),
                                           (2160, 3840, 8)])        # C4's frame (BASELINE configs[3]): what bench.py --gpus N gathers in its `c4` block
def test_two_rank_gather_reassembles_image(h, w, tile_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, h, w, tile_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_covers_every_row_once():
    for h in (1, 7, 8, 9, 135, 1080, 2160):
        for nranks in (1, 2, 3, 4, 8):
            for tr in (8, 16):
                seen = np.concatenate([api.local_rows_of(h, r, nranks, tr) for r in range(nranks)])
                assert sorted(seen.tolist()) == list(range(h))


def test_assemble_numpy():
    h, w = 37, 6
    full = np.arange(h * w, dtype=np.uint32).reshape(h, w)
    parts = [full[api.local_rows_of(h, r, 3, 8)] for r in range(3)]
    assert np.array_equal(rdist.assemble_numpy(parts, h, w, 3, 8), full.reshape(-1))


@pytest.mark.parametrize("h,w,n,tr", [(1080, 16, 8, 8), (1080, 16, 4, 16), (135, 7, 8, 8), (50, 33, 3, 8), (17, 5, 2, 16),
                                       (8, 8, 2, 8), (7, 3, 4, 8), (2160, 4, 8, 8), (100, 9, 7, 8), (64, 5, 1, 8)])
def test_strided_reassembly_equals_the_index_permutation(h, w, n, tr):
    """FrameGatherer._assemble (at most three strided copies) against the row-index permutation and
    against the image itself, with the blocks filled as the ranks would send them (no process group)."""
    full = (np.arange(h * w, dtype=np.int64) * 2654435761 % (1 << 31)).astype(np.int32).reshape(h, w)
    g = rdist.FrameGatherer(h, w, 0, n, tr, "cpu", slots=2)
    for k in (0, 1):
        g._stacked[k].fill_(-1)
        for r in range(n):
            rows = api.local_rows_of(h, r, n, tr)
            g._stacked[k][r, : len(rows)] = torch.from_numpy(full[rows] + k)
        out = g._assemble(k)
        assert np.array_equal(out.numpy(), full + k)
        assert torch.equal(out, g._assemble_by_index(k))
