"""The N>1 path of bench.py on CPU (no GPU, gloo): the slice arithmetic of the
two scaling modes, the seeded chunk list (a rank's slice is the same bytes
whatever the world size), the max-time / sum-bytes reduction, and the
self-starting launcher (`python bench.py --gpus 2` with no WORLD_SIZE)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_slice(2000, world, rank)
    local_bytes = (hi - lo) * bench.CHUNK
    local_wall = 0.5 + 0.25 * rank          # rank 1 is the slow one
    dist.barrier()
    value, wall = bench.aggregate_throughput(local_wall, local_bytes, steps=5, dist_mod=dist, device="cpu")
    q.put((rank, (lo, hi), value, wall))
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
    assert s0 == (0, 1000) and s1 == (1000, 2000)     # contiguous slices of one list
    assert w0 == w1 == pytest.approx(0.75)            # MAX over ranks
    want = 2 * 1000 * 65536 / (0.75 / 5) / 1e9        # all ranks' bytes over the slowest rank's time per step
    assert v0 == pytest.approx(want) and v1 == pytest.approx(want)


def test_single_rank_needs_no_process_group():
    import bench
    v, w = bench.aggregate_throughput(2.0, 10 * 65536, steps=4)
    assert w == 2.0 and v == pytest.approx(10 * 65536 / 0.5 / 1e9)


def test_slices_cover_the_list_exactly():
    import bench
    for total, world in ((163840, 8), (163840, 4), (163840, 1), (100001, 8), (7, 8), (10, 3), (0, 2)):
        got = [bench.shard_slice(total, world, r) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))                 # contiguous, no overlap
        per = (total + world - 1) // world
        assert all(hi - lo <= per for lo, hi in got)
    assert bench.shard_slice(163840, 8, 3) == (61440, 81920)                   # BASELINE config 5: 20 480 per GPU


def test_a_slice_holds_the_same_bytes_in_any_world():
    import bench
    whole = bench.gen_data("harness", 4090, 4110, "cpu", 0x5EED0003)           # crosses a generator block
    for world in (2, 4):
        parts = []
        for r in range(world):
            lo, hi = bench.shard_slice(20, world, r)
            parts.append(bench.gen_data("harness", 4090 + lo, 4090 + hi, "cpu", 0x5EED0003))
        assert torch.equal(torch.cat(parts), whole)
    assert not torch.equal(whole[:bench.CHUNK], whole[bench.CHUNK:2 * bench.CHUNK])


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--chunks", "1000",
                        "--steps", "4"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["slowest_rank_wall_s"] == pytest.approx(2.0)
    assert out["weak_bytes_per_step_GB"] == pytest.approx(2 * 1000 * 65536 / 1e9)
    assert out["strong_bytes_per_step_GB"] == pytest.approx(163840 * 65536 / 1e9)
