"""Sharding of independent IQ captures over ranks.

One process per GPU; stream s belongs to rank s // per_rank (contiguous blocks,
so that a rank's captures are adjacent in its HBM buffer).  The only collective
on this path is the broadcast of the shared taps from rank 0 at set-up; there
is no data-path exchange.  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.
"""
import numpy as np


def shard_streams(n_streams, rank, world):
    """stream ids owned by `rank` when n_streams captures are split over `world`
    ranks as evenly as possible (first n_streams % world ranks get one more)."""
    base, rem = divmod(int(n_streams), int(world))
    start = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return list(range(start, start + count))


def broadcast_taps(taps, dist=None, device=None, src=0):
    """rank `src` supplies the complex taps; every rank returns the same
    complex64 array.  Without an initialised process group this is the identity."""
    import torch
    t = np.ascontiguousarray(taps, dtype=np.complex64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    n = torch.tensor([len(t)], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = torch.zeros((int(n.item()), 2), dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        buf.copy_(torch.from_numpy(t.view(np.float32).reshape(-1, 2)))
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy().reshape(-1).view(np.complex64).copy()


def max_over_ranks(value, dist=None, device=None):
    """the benchmark contract: elapsed time is the MAX over ranks"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
