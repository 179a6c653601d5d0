"""CPU, world_size 2 over gloo: the multi-GPU path's host logic — block-aligned time chunks, counter
seeds from the closed form, ordered point-to-point gather — with the oracle standing in for the kernel
(there is no GPU here; the kernel itself is covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from doppler_amd import shard
    from oracle import oracle as orc
    shift, rate = 815000.0, 2400000
    n = 2048 * 21 + 333
    rng = np.random.default_rng(99)
    x = rng.integers(-20000, 20000, size=2 * n, dtype=np.int16)          # same stream on every rank
    lo, hi = shard.chunk_bounds(n, world, rank)
    seed = shard.chunk_seed(shift, rate, lo)
    y, _ = orc.const_stream(x[2 * lo:2 * hi], "i16", "i16", int(shift), rate, samplenum=seed)
    sizes = []
    for r in range(world):
        a, b = shard.chunk_bounds(n, world, r)
        sizes.append(2 * (b - a))
    full = shard.ordered_gather(torch.from_numpy(y.view(np.int16).copy()), sizes, dst=0)
    if rank == 0:
        want, _ = orc.const_stream(x, "i16", "i16", int(shift), rate)
        q.put(bool(np.array_equal(full.numpy().view(np.uint8), want)))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_time_chunks_and_ordered_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
