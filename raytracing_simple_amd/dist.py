"""Multi-GPU frame assembly: one process per GPU, image sharded by interleaved row tiles
(SURVEY 8e), ONE collective per frame -- a gather of the packed uint32 rows to rank 0 over
RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  No other
data-path communication exists: pixels are independent."""
import numpy as np
import torch
import torch.distributed as dist

from .api import local_rows_of


def max_local_rows(h, nranks, tile_rows):
    return max(len(local_rows_of(h, r, nranks, tile_rows)) for r in range(nranks))


def scatter_index(h, w, nranks, tile_rows, device="cpu"):
    """Row permutation that turns the concatenation of all ranks' padded local buffers into the
    image: full[row_of[k]] = gathered[k] for the valid slots."""
    pad = max_local_rows(h, nranks, tile_rows)
    src, dst = [], []
    for r in range(nranks):
        rows = local_rows_of(h, r, nranks, tile_rows)
        src.extend(r * pad + i for i in range(len(rows)))
        dst.extend(int(v) for v in rows)
    return (torch.tensor(src, dtype=torch.long, device=device),
            torch.tensor(dst, dtype=torch.long, device=device), pad)


class FrameGatherer:
    """Pre-allocated buffers for the per-frame gather.

    `slots` send buffers (int32 [pad_rows, w], rows beyond the rank's own count are padding) let
    the collective of frame k run while frame k+1 renders into the next slot: the renderer
    writes straight into `local(k)` (rt_set_pixel_buffer), `gather(k, async_op=True)` queues the
    collective behind that launch and returns at once, `wait(k)` is only needed before slot k
    is rendered into again or its frame is read."""

    def __init__(self, h, w, rank, nranks, tile_rows, device, group=None, dst=0, slots=1):
        self.h, self.w, self.rank, self.nranks, self.dst = h, w, rank, nranks, dst
        self.group = group
        self.src_idx, self.dst_idx, self.pad = scatter_index(h, w, nranks, tile_rows, device)
        self.n_local = len(local_rows_of(h, rank, nranks, tile_rows))
        self.slots = slots
        self._local = [torch.zeros((self.pad, w), dtype=torch.int32, device=device) for _ in range(slots)]
        self._work = [None] * slots
        self._parts = None
        self._full = None
        self.tile_rows = tile_rows
        if rank == dst:
            # one contiguous [nranks, pad, w] block per slot; the collective writes into its rows
            self._stacked = [torch.zeros((nranks, self.pad, w), dtype=torch.int32, device=device) for _ in range(slots)]
            self._parts = [list(st.unbind(0)) for st in self._stacked]
            self._full = [torch.zeros((h, w), dtype=torch.int32, device=device) for _ in range(slots)]

    # single-slot conveniences (tests, simple callers)
    @property
    def local(self):
        return self._local[0]

    @property
    def full(self):
        return self._full[0] if self._full else None

    def local_slot(self, k):
        return self._local[k % self.slots]

    def gather(self, k=0, async_op=False):
        """The frame-end collective of slot k.  Returns the assembled [h, w] image on rank dst
        (None elsewhere); with async_op=True returns None immediately -- call wait(k) later."""
        k %= self.slots
        if self.nranks == 1:
            self._full[k].copy_(self._local[k][: self.h])
            return self._full[k]
        work = dist.gather(self._local[k], self._parts[k] if self.rank == self.dst else None, dst=self.dst,
                           group=self.group, async_op=async_op)
        if async_op:
            self._work[k] = work
            return None
        return self._assemble(k)

    def wait(self, k=0):
        """Complete an async gather of slot k and assemble the frame (rank dst)."""
        k %= self.slots
        if self._work[k] is not None:
            self._work[k].wait()
            self._work[k] = None
            return self._assemble(k)
        return self._full[k] if self._full else None

    def _assemble(self, k):
        """De-interleave the gathered blocks into the image on the gather root: the library's HIP kernel
        (rt_deinterleave_rows, include/rt_api.h) on the current stream for device buffers.  Host tensors
        (the gloo rehearsals of the CPU test-suite, which have no device) go through _assemble_strided."""
        if self.rank != self.dst:
            return None
        st, full = self._stacked[k], self._full[k]
        if st.is_cuda:
            from . import api
            api.deinterleave_rows(full.data_ptr(), st.data_ptr(), self.w, self.h, self.nranks, self.tile_rows, self.pad,
                                  device=st.device.index or 0, stream=torch.cuda.current_stream(st.device).cuda_stream)
            return full
        return self._assemble_strided(k)

    def _assemble_strided(self, k):
        """The same permutation with at most three strided copies (tile t of the image is local tile
        t // n of rank t % n): all complete groups of n full-height tiles at once, the last incomplete
        group, the short tile at the bottom.  Host tensors, and the cross-check of the kernel in tests."""
        if self.rank != self.dst:
            return None
        n, tr, w, h = self.nranks, self.tile_rows, self.w, self.h
        st, full = self._stacked[k], self._full[k]
        t_full = h // tr                     # full-height tiles
        g = t_full // n                      # complete groups of n tiles
        if g:
            full[: g * n * tr].view(g, n, tr * w).copy_(st[:, : g * tr].reshape(n, g, tr * w).transpose(0, 1))
        r = t_full - g * n                   # full-height tiles of the last, incomplete group: ranks 0..r-1
        if r:
            full[g * n * tr:(g * n + r) * tr].view(r, tr * w).copy_(st[:r, g * tr:(g + 1) * tr].reshape(r, tr * w))
        short = h - t_full * tr              # rows of the short tile, owned by rank t_full % n
        if short:
            owner, j = t_full % n, t_full // n
            full[t_full * tr:].copy_(st[owner, j * tr: j * tr + short])
        return full

    def _assemble_by_index(self, k):
        """The same permutation through the row index tables (kept as the check of _assemble)."""
        stacked = self._stacked[k].view(self.nranks * self.pad, self.w)
        out = torch.zeros_like(self._full[k])
        out.index_copy_(0, self.dst_idx, stacked.index_select(0, self.src_idx))
        return out


def assemble_numpy(parts, h, w, nranks, tile_rows):
    """Host-side de-interleave of per-rank local row blocks (lists of uint32 arrays)."""
    full = np.zeros((h, w), np.uint32)
    for r, part in enumerate(parts):
        rows = local_rows_of(h, r, nranks, tile_rows)
        full[rows] = np.asarray(part, dtype=np.uint32).reshape(-1, w)[: len(rows)]
    return full.reshape(-1)
