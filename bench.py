#!/usr/bin/env python3
"""bench.py -- batched LZ4 compress + decompress throughput on MI355X.

Workload (BASELINE.json configs[1]): 100 000 chunks x 64 KiB of synthetic int32
data per GPU, through hipcompBatchedLZ4CompressAsync +
hipcompBatchedLZ4DecompressAsync of hipcomp-core_amd/lib/libhipcomp.so.  One
"step" = one compress pass + one decompress pass over the whole chunk list,
inputs resident in HBM.  value = uncompressed GB (1e9) through the round trip
per second, summed over ranks (chunks are sharded, no collective: "weak").

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with
`roofline` (dominant kernel = the compress kernel, HBM-bound model) and
`cpu_baseline` (system liblz4 on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 65536
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def gen_data(dist: str, n_chunks: int, device, seed: int) -> torch.Tensor:
    """n_chunks x 64 KiB of int32 values, generated on the device (seeded)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_ints = n_chunks * (CHUNK // 4)
    out = torch.empty(n_ints, dtype=torch.int32, device=device)
    piece = 1 << 26
    for off in range(0, n_ints, piece):
        m = min(piece, n_ints - off)
        if dist == "uniform":      # incompressible
            out[off:off + m] = torch.randint(-(1 << 31), (1 << 31) - 1, (m,), dtype=torch.int64, device=device, generator=g).to(torch.int32)
        elif dist == "harness":    # 300 + (x & 3)  
            out[off:off + m] = torch.randint(300, 304, (m,), dtype=torch.int32, device=device, generator=g)
        elif dist == "runs":       # value = run index, run length U[1,16]
            lens = torch.randint(1, 17, (m // 8 + 16,), dtype=torch.int64, device=device, generator=g)
            vals = torch.repeat_interleave(torch.arange(lens.numel(), dtype=torch.int32, device=device), lens)
            while vals.numel() < m:
                vals = torch.cat([vals, vals + vals[-1] + 1])
            out[off:off + m] = vals[:m]
        else:
            raise ValueError(dist)
    return out.view(torch.uint8)


class Lz4Job:
    """Buffers of one rank, allocated once like a caller of the C API would."""

    def __init__(self, hc, lib, data: torch.Tensor, dtype: int):
        self.hc = hc
        self.codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=lib)
        self.src = hc.batch.from_device_buffer(data, CHUNK)
        n = self.src.n
        dev = data.device
        self.n = n
        self.comp = hc.batch.alloc_batch(n, self.codec.max_output_chunk_size(CHUNK), dev)
        self.temp = torch.empty(max(self.codec.compress_temp_size(n, CHUNK), 8), dtype=torch.uint8, device=dev)
        self.dtemp = torch.empty(max(self.codec.decompress_temp_size(n, CHUNK), 8), dtype=torch.uint8, device=dev)
        self.out = hc.batch.alloc_batch(n, CHUNK, dev)
        self.caps = torch.full((n,), CHUNK, dtype=torch.int64, device=dev)
        self.actual = torch.zeros(n, dtype=torch.int64, device=dev)
        self.statuses = torch.zeros(n, dtype=torch.int32, device=dev)

    def compress(self):
        st = self.codec.compress_async(self.src, CHUNK, self.temp, self.comp)
        assert st == 0, st

    def decompress(self):
        st = self.codec.decompress_async(self.comp, self.caps, self.actual, self.dtemp, self.out, self.statuses)
        assert st == 0, st

    def verify(self):
        assert int(self.statuses.abs().sum().item()) == 0, "decompress reported errors"
        assert bool((self.actual == self.src.sizes).all().item())
        a = self.out.data[: self.n * CHUNK].view(torch.int64)
        b = self.src.data[: self.n * CHUNK].view(torch.int64)
        assert bool(torch.equal(a, b)), "round trip mismatch"


def time_phases(job: Lz4Job, steps: int):
    """Per-step HIP-event times (ms) of the two kernels, on the launch stream."""
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    for e in evs:
        e[0].record()
        job.compress()
        e[1].record()
        job.decompress()
        e[2].record()
    torch.cuda.synchronize()
    tc = [e[0].elapsed_time(e[1]) for e in evs]
    td = [e[1].elapsed_time(e[2]) for e in evs]
    return tc, td


def cpu_liblz4_baseline(sample: np.ndarray, reps: int = 3):
    """System liblz4 round trip on the host cores (BASELINE.json configs[0]):
    oracle/cpu_baseline.c -- pthreads, static contiguous partition, best of
    `reps`.  Falls back to the scalar C restatement when liblz4 is absent."""
    from oracle import oracle as O
    L = O.lib()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = sample.size // CHUNK
    tc, td, ct = ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t()
    L.cpu_liblz4_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_size_t)]
    rc = L.cpu_liblz4_roundtrip(sample.ctypes.data, n, CHUNK, cores, reps, ctypes.byref(tc), ctypes.byref(td),
                                ctypes.byref(ct))
    total = n * CHUNK
    if rc == 0:
        return {
            "value": total / (tc.value + td.value) / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"system liblz4 (LZ4_compress_default + LZ4_decompress_safe via oracle/cpu_baseline.c), "
                      f"{n} x 64KiB chunks of the same data, {cores} pthreads, best of {reps}; "
                      f"compress {total / tc.value / 1e9:.2f} GB/s, decompress {total / td.value / 1e9:.2f} GB/s, "
                      f"ratio {total / max(ct.value, 1):.3f}",
        }
    # no liblz4 on this box: time the C restatement (scalar, 1 core)
    t1 = time.perf_counter()
    kk = min(64, n)
    for i in range(kk):
        O.lz4_compress(bytes(sample[i * CHUNK:(i + 1) * CHUNK]), 1, CHUNK)
    dt = time.perf_counter() - t1
    return {"value": kk * CHUNK / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle/lz4_oracle.c compress only, {kk} chunks (liblz4 unavailable, rc={rc})"}


# ---- multi-GPU plumbing (chunks shard with no data-path collective) ----------

def shard_seed(base_seed: int, rank: int) -> int:
    """Every rank owns its own slice of the chunk list: same size, own seed."""
    return base_seed + rank


def aggregate_throughput(local_wall_s: float, local_bytes: int, steps: int, dist_mod=None, device=None):
    """Whole-job GB/s: all ranks' bytes over the slowest rank's time."""
    wall, total = local_wall_s, float(local_bytes)
    if dist_mod is not None and dist_mod.is_initialized() and dist_mod.get_world_size() > 1:
        if dist_mod.get_backend() != "nccl":
            device = "cpu"
        t = torch.tensor([local_wall_s], dtype=torch.float64, device=device)
        b = torch.tensor([float(local_bytes)], dtype=torch.float64, device=device)
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        dist_mod.all_reduce(b, op=dist_mod.ReduceOp.SUM)
        wall, total = float(t.item()), float(b.item())
    return total / (wall / steps) / 1e9, wall


def measure_variant(hc, lib, dev, dist_name: str, dtype_name: str, chunks: int, seed: int):
    """Compress / decompress ms of one (distribution, data_type) row."""
    data = gen_data(dist_name, chunks, dev, seed)
    dtype = hc.hipcompType.CHAR if dtype_name == "char" else hc.hipcompType.INT
    job = Lz4Job(hc, lib, data, dtype)
    job.compress(); job.decompress(); torch.cuda.synchronize()
    job.verify()
    tc, td = time_phases(job, 2)
    nb, cb = job.n * CHUNK, int(job.comp.sizes.sum().item())
    return {"distribution": dist_name, "data_type": dtype_name.upper(), "chunks": chunks, "ratio": nb / max(cb, 1),
            "compress_GBps": nb / (min(tc) * 1e-3) / 1e9, "decompress_GBps": nb / (min(td) * 1e-3) / 1e9,
            "roundtrip_GBps": nb / ((min(tc) + min(td)) * 1e-3) / 1e9,
            "hbm_frac_compress": (nb + cb) / (min(tc) * 1e-3) / 1e9 / HBM_PEAK_GBS}


def pmc_traffic(args):
    """HBM bytes per compress launch from the committed rocprofv3 PMC passes
    (profiles/lz4_hbm_traffic.json: FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes, + WRITE_SIZE), or None when no pass exists
    for this workload."""
    path = os.path.join(ROOT, "profiles", "lz4_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    key = f"{args.dist}/{args.dtype}/{args.chunks}"
    return table.get(key, {}).get("traffic_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=100000, help="chunks per GPU")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "harness", "runs"])
    ap.add_argument("--dtype", default="char", choices=["char", "int"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ref", action="store_true", help="also time the reference build (oracle/_ref) on the same buffers")
    ap.add_argument("--cpu-sample-chunks", type=int, default=16384)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier / reductions "
                    "(nccl = RCCL; gloo only for rehearsing the N>1 path on a one-GPU box)")
    ap.add_argument("--no-variants", dest="variants", action="store_false",
                    help="skip the extra (distribution, data_type) rows")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    local = local % max(torch.cuda.device_count(), 1)  # (rehearsal: several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    hc = importlib.import_module("hipcomp-core_amd")
    lib = hc.default_library()
    dtype = hc.hipcompType.CHAR if args.dtype == "char" else hc.hipcompType.INT

    data = gen_data(args.dist, args.chunks, dev, seed=shard_seed(0x5EED0002, rank))
    job = Lz4Job(hc, lib, data, dtype)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.compress()
        job.decompress()
    torch.cuda.synchronize()
    job.verify()

    barrier()
    t0 = time.perf_counter()
    tc, td = time_phases(job, args.steps)
    barrier()
    wall_local = time.perf_counter() - t0
    n_bytes = job.n * CHUNK
    c_bytes = int(job.comp.sizes.sum().item())
    value, wall = aggregate_throughput(wall_local, n_bytes, args.steps, dist if dist_on else None, dev)
    ms_step = wall / args.steps * 1e3

    if rank == 0:
        tc_avg, td_avg = sum(tc) / len(tc), sum(td) / len(td)
        algo = n_bytes + c_bytes  # per launch: N read + C written
        res = {
            "metric": "LZ4 batched compress+decompress GB/s, 64KiB chunks",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if args.dtype == "char" else "u32", "data": "synthetic",
            "config": {
                "workload": f"hipcompBatchedLZ4 compress+decompress, {args.chunks}x64KiB int32 chunks per GPU, "
                            f"{args.dist} data, data_type={'CHAR' if args.dtype == 'char' else 'INT'}",
                "chunks_per_gpu": args.chunks, "chunk_bytes": CHUNK, "distribution": args.dist,
                "parallelism": f"{world} independent chunk shards, no collective",
            },
            "ratio": n_bytes / max(c_bytes, 1),
            "compress_GBps": n_bytes / (tc_avg * 1e-3) / 1e9,
            "decompress_GBps": n_bytes / (td_avg * 1e-3) / 1e9,
            "compress_ms": tc_avg, "decompress_ms": td_avg,
            "roofline": {
                "bound": "hbm", "kernel": "lz4_compress_kernel",
                "achieved": algo / (tc_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": algo / (tc_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(args),
                "algorithmic_bytes_per_launch": algo,
                "decompress_achieved": algo / (td_avg * 1e-3) / 1e9,
            },
        }
        if args.ref:
            from oracle import oracle as O
            if os.path.exists(O.REF_LIB_PATH):
                rjob = Lz4Job(hc, hc.HipcompLibrary(O.REF_LIB_PATH), data, dtype)
                rjob.compress(); rjob.decompress(); torch.cuda.synchronize()
                rtc, rtd = time_phases(rjob, 2)
                same = bool(torch.equal(rjob.comp.sizes, job.comp.sizes))
                res["reference_gpu"] = {
                    "note": "reference's own kernels (oracle/_ref) on the same GPU and buffers",
                    "compress_ms": min(rtc), "decompress_ms": min(rtd),
                    "roundtrip_GBps": n_bytes / ((min(rtc) + min(rtd)) * 1e-3) / 1e9,
                    "compressed_sizes_identical": same,
                }
                del rjob
        if args.variants and world == 1:
            res["variants"] = [measure_variant(hc, lib, dev, dn, tn, min(args.chunks, 20000), 0x5EED0003 + i)
                               for i, (dn, tn) in enumerate((("uniform", "int"), ("harness", "char"), ("runs", "char")))]
        if not args.no_cpu and world == 1:  # reported at N=1 only
            k = min(args.cpu_sample_chunks, job.n)
            res["cpu_baseline"] = cpu_liblz4_baseline(data[: k * CHUNK].cpu().numpy())
        print(json.dumps(res))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
