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
    """Pre-allocated buffers for the per-frame gather.  `local` is this rank's padded buffer:
    int32 [pad_rows, w]; rows beyond the rank's own count are padding."""

    def __init__(self, h, w, rank, nranks, tile_rows, device, group=None, dst=0):
        self.h, self.w, self.rank, self.nranks, self.dst = h, w, rank, nranks, dst
        self.group = group
        self.src_idx, self.dst_idx, self.pad = scatter_index(h, w, nranks, tile_rows, device)
        self.local = torch.zeros((self.pad, w), dtype=torch.int32, device=device)
        self.n_local = len(local_rows_of(h, rank, nranks, tile_rows))
        self.parts = None
        self.full = None
        if rank == dst:
            self.parts = [torch.zeros((self.pad, w), dtype=torch.int32, device=device)
                          for _ in range(nranks)]
            self.full = torch.zeros((h, w), dtype=torch.int32, device=device)

    def gather(self):
        """The frame-end collective.  Returns the assembled [h, w] image on rank dst, else None."""
        if self.nranks == 1:
            self.full.copy_(self.local[: self.h])
            return self.full
        dist.gather(self.local, self.parts if self.rank == self.dst else None, dst=self.dst,
                    group=self.group)
        if self.rank != self.dst:
            return None
        stacked = torch.cat(self.parts, dim=0)
        self.full.index_copy_(0, self.dst_idx, stacked.index_select(0, self.src_idx))
        return self.full


def assemble_numpy(parts, h, w, nranks, tile_rows):
    """Host-side de-interleave of per-rank local row blocks (lists of uint32 arrays)."""
    full = np.zeros((h, w), np.uint32)
    for r, part in enumerate(parts):
        rows = local_rows_of(h, r, nranks, tile_rows)
        full[rows] = np.asarray(part, dtype=np.uint32).reshape(-1, w)[: len(rows)]
    return full.reshape(-1)
