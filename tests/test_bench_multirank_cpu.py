"""The N>1 path of bench.py on CPU: two gloo ranks, each with its own shard,
no data-path collective; only the barrier and the max-time / sum-bytes
reduction that turn per-rank numbers into the whole-job value."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank-private shard: same size, different seed
    seed = bench.shard_seed(0x5EED0002, rank)
    local_bytes = 1000 * bench.CHUNK
    local_wall = 0.5 + 0.25 * rank          # rank 1 is the slow one
    dist.barrier()
    value, wall = bench.aggregate_throughput(local_wall, local_bytes, steps=5, dist_mod=dist, device="cpu")
    q.put((rank, seed, value, wall))
    dist.destroy_process_group()


def test_two_rank_aggregation_with_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, s0, v0, w0), (r1, s1, v1, w1) = out
    assert s0 != s1                                   # each rank has its own shard
    assert w0 == w1 == pytest.approx(0.75)            # MAX over ranks
    want = 2 * 1000 * 65536 / (0.75 / 5) / 1e9        # all ranks' bytes over the slowest rank's time per step
    assert v0 == pytest.approx(want) and v1 == pytest.approx(want)


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    v, w = bench.aggregate_throughput(2.0, 10 * 65536, steps=4)
    assert w == 2.0 and v == pytest.approx(10 * 65536 / 0.5 / 1e9)
