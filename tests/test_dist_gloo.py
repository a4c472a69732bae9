"""CPU test of the N>1 path with world_size 2 on gloo: capture sharding, the
taps broadcast (the path's only collective) and the max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GRHIP_NO_TORCH_PRELOAD"] = "1"
    import torch.distributed as dist
    import grhip_loader
    g = grhip_loader.import_grhip()
    from grhip import dist as gd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = g.workload
    taps = wl.cfg2_proto_taps() if rank == 0 else np.zeros(3, np.complex64)    # only rank 0 has them
    got = gd.broadcast_taps(taps, dist)
    mine = gd.shard_streams(5, rank, world)
    t = gd.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, got.tobytes(), mine, t))


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in ps])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    sys.path.insert(0, ROOT)
    import grhip_loader
    ref = grhip_loader.import_grhip().workload.cfg2_proto_taps().tobytes()
    assert res[0][1] == ref and res[1][1] == ref            # every rank got rank 0's taps
    assert res[0][2] == [0, 1, 2] and res[1][2] == [3, 4]   # disjoint, complete, contiguous
    assert res[0][3] == res[1][3] == 2.0                    # MAX over ranks


@pytest.mark.parametrize("n,world", [(8, 8), (8, 3), (1, 4), (0, 2), (17, 4)])
def test_shard_streams_partition(n, world):
    sys.path.insert(0, ROOT)
    import grhip_loader
    grhip_loader.import_grhip()
    from grhip import dist as gd
    parts = [gd.shard_streams(n, r, world) for r in range(world)]
    flat = [s for p in parts for s in p]
    assert flat == list(range(n))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
