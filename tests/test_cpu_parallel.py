"""N>1 path on CPU: world_size-2 gloo run of the pair partition + strip gather (surround360_amd/parallel.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from surround360_amd import parallel


def test_partition_pairs():
    assert parallel.partition_pairs(14, 8) == [0, 2, 4, 6, 8, 10, 12, 13, 14]
    assert parallel.partition_pairs(14, 1) == [0, 14]
    assert parallel.partition_pairs(14, 4) == [0, 4, 8, 11, 14]
    b = parallel.partition_pairs(14, 16)  # more ranks than pairs: trailing ranks get nothing
    assert b[-1] == 14 and all(0 <= b[i + 1] - b[i] <= 1 for i in range(16))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, camH, stripW = 14, 6, 5
    bounds = parallel.partition_pairs(P, world)
    strips = torch.zeros((2, P, camH, stripW, 4), dtype=torch.uint8)
    for p in range(bounds[rank], bounds[rank + 1]):
        for eye in range(2):
            strips[eye, p] = 10 * p + eye + 1  # what this rank "rendered"
    parallel.gather_strips(strips, bounds, rank, world, 0)
    if rank == 0:
        exp = torch.zeros_like(strips)
        for p in range(P):
            for eye in range(2):
                exp[eye, p] = 10 * p + eye + 1
        q.put(bool(torch.equal(strips, exp)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_strips_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
