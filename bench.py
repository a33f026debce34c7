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
import threading
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


def cpu_liblz4_baseline(sample: np.ndarray, budget_s: float = 12.0):
    """System liblz4 round trip on the host cores (BASELINE.json configs[0])."""
    try:
        lz4 = ctypes.CDLL("liblz4.so.1")
    except OSError:
        return None
    lz4.LZ4_compressBound.argtypes = [ctypes.c_int]
    lz4.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lz4.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = sample.size // CHUNK
    bound = lz4.LZ4_compressBound(CHUNK)
    comp = np.empty(n * bound, dtype=np.uint8)
    out = np.empty(n * CHUNK, dtype=np.uint8)
    csz = np.zeros(n, dtype=np.int64)
    sp, cp, op = sample.ctypes.data, comp.ctypes.data, out.ctypes.data

    def work(lo, hi, phase):
        for i in range(lo, hi):
            if phase == 0:
                csz[i] = lz4.LZ4_compress_default(sp + i * CHUNK, cp + i * bound, CHUNK, bound)
            else:
                lz4.LZ4_decompress_safe(cp + i * bound, op + i * CHUNK, int(csz[i]), CHUNK)

    def run(phase):
        per = (n + cores - 1) // cores
        ts = [threading.Thread(target=work, args=(k * per, min(n, (k + 1) * per), phase)) for k in range(cores)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return time.perf_counter() - t0

    best_c = best_d = 1e30
    t_start = time.perf_counter()
    reps = 0
    while reps < 3 and time.perf_counter() - t_start < budget_s:
        best_c = min(best_c, run(0))
        best_d = min(best_d, run(1))
        reps += 1
    assert bytes(out[:CHUNK]) == bytes(sample[:CHUNK])
    total = n * CHUNK
    return {
        "value": total / (best_c + best_d) / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
        "sample": f"system liblz4 (LZ4_compress_default + LZ4_decompress_safe), {n} x 64KiB chunks of the same data, "
                  f"{cores} threads, best of {reps}; compress {total / best_c / 1e9:.2f} GB/s, "
                  f"decompress {total / best_d / 1e9:.2f} GB/s, ratio {total / max(int(csz.sum()), 1):.3f}",
    }


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
    ap.add_argument("--cpu-sample-chunks", type=int, default=8192)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    hc = importlib.import_module("hipcomp-core_amd")
    lib = hc.default_library()
    dtype = hc.hipcompType.CHAR if args.dtype == "char" else hc.hipcompType.INT

    data = gen_data(args.dist, args.chunks, dev, seed=0x5EED0002 + rank)
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
    wall = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    n_bytes = job.n * CHUNK
    c_bytes = int(job.comp.sizes.sum().item())
    ms_step = wall / args.steps * 1e3
    value = world * n_bytes / (wall / args.steps) / 1e9

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
                "frac": algo / (tc_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
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
        if not args.no_cpu:
            k = min(args.cpu_sample_chunks, job.n)
            sample = data[: k * CHUNK].cpu().numpy()
            cb = cpu_liblz4_baseline(sample)
            if cb is None:
                # no liblz4 on this box: time the C restatement (scalar, 1 core)
                from oracle import oracle as O
                t1 = time.perf_counter()
                kk = min(64, k)
                for i in range(kk):
                    O.lz4_compress(bytes(sample[i * CHUNK:(i + 1) * CHUNK]), 1, CHUNK)
                dt = time.perf_counter() - t1
                cb = {"value": kk * CHUNK / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                      "sample": f"oracle/lz4_oracle.c compress only, {kk} chunks (liblz4 unavailable)"}
            res["cpu_baseline"] = cb
        print(json.dumps(res))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
