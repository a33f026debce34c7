#!/usr/bin/env python3
"""bench.py -- batched LZ4 compress + decompress throughput on MI355X.

Workload (BASELINE.json configs[1]): 100 000 chunks x 64 KiB of synthetic int32
data per GPU, through hipcompBatchedLZ4CompressAsync +
hipcompBatchedLZ4DecompressAsync of hipcomp-core_amd/lib/libhipcomp.so.  One
"step" = one compress pass + one decompress pass over the rank's chunks,
inputs resident in HBM.  value = uncompressed GB (1e9) through the round trip
per second, summed over ranks.

Multi-GPU (BASELINE.json configs[4]): chunks are independent, so
rank r owns a contiguous slice of ONE seeded chunk list and no collective sits
on the data path (no RCCL: the start/stop barrier and the two scalar
reductions run over gloo).  Two modes, both in the JSON line when N > 1 or
--config5 is given:
  weak   (the headline `value`): --chunks per GPU, the list has N x chunks;
  strong (`config5_strong`):     --total-chunks (default 163 840 = 10 GiB) in
                                 all, rank r gets [r*ceil(B/N), (r+1)*ceil(B/N)).

    python bench.py [--gpus N --steps K --warmup W]
`--gpus N` with N > 1 needs no launcher: when WORLD_SIZE is not set the process
starts N child ranks itself, before anything touches the GPU.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it
is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with
`roofline` (dominant kernel = the LZ4 compress kernel, HBM-bound model),
`cpu_baseline` (system liblz4 on the host cores, bounded sample) and, at N = 1,
`extra_keys`: every other BASELINE config (LZ4 distributions x data types at
the full 100 000 chunks, Snappy on TPC-H-like text, Cascaded on sorted
columns), each with its own HBM fraction, ratio and CPU baseline.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 65536
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
GEN_BLOCK = 4096       # chunks per generator block: block b of the list is seeded with seed + b
CONFIG5_TOTAL = 163840  # 10 GiB of 64 KiB chunks (BASELINE.json configs[4])


# ---- chunk-list arithmetic (pure Python: covered by the CPU tests) -----------

def shard_slice(total_chunks: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of a list of `total_chunks` for `rank`:
    ceil(B/G) chunks per rank, the last ranks may get fewer."""
    per = (total_chunks + world - 1) // world
    lo = min(rank * per, total_chunks)
    return lo, min(lo + per, total_chunks)


def gen_blocks(lo: int, hi: int):
    """Generator blocks that cover chunks [lo, hi): (block, first, last) with
    first/last relative to the block."""
    out = []
    b = lo // GEN_BLOCK
    while b * GEN_BLOCK < hi:
        first = max(lo, b * GEN_BLOCK) - b * GEN_BLOCK
        last = min(hi, (b + 1) * GEN_BLOCK) - b * GEN_BLOCK
        out.append((b, first, last))
        b += 1
    return out


def self_spawn(argv, n: int) -> int:
    """Start n child ranks of this script (before this process has touched the
    GPU) and wait for them; rank 0 prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def aggregate_throughput(local_wall_s: float, local_bytes: int, steps: int, dist_mod=None, device=None):
    """Whole-job GB/s: all ranks' bytes over the slowest rank's time."""
    import torch
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


# ---- synthetic inputs, generated on the device ----------------

def _gen_block(dist: str, block: int, device, seed: int):
    """GEN_BLOCK chunks x 64 KiB of int32 values: block `block` of the list `dist`/`seed`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed + block)
    m = GEN_BLOCK * (CHUNK // 4)
    if dist == "uniform":      # incompressible (8d 2a)
        return torch.randint(-(1 << 31), (1 << 31) - 1, (m,), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    if dist == "harness":      # 300 + (x & 3)  (8d 2b)
        return torch.randint(300, 304, (m,), dtype=torch.int32, device=device, generator=g)
    if dist == "runs":         # value = run index, run length U[1,16] (8d 2c)
        lens = torch.randint(1, 17, (m // 8 + 16,), dtype=torch.int64, device=device, generator=g)
        vals = torch.repeat_interleave(torch.arange(lens.numel(), dtype=torch.int32, device=device), lens)
        while vals.numel() < m:
            vals = torch.cat([vals, vals + vals[-1] + 1])
        return vals[:m].contiguous()
    raise ValueError(dist)


def gen_data(dist: str, lo: int, hi: int, device, seed: int):
    """Chunks [lo, hi) of the seeded list, as one uint8 device buffer."""
    import torch
    out = torch.empty((hi - lo) * (CHUNK // 4), dtype=torch.int32, device=device)
    at = 0
    per = CHUNK // 4
    for b, first, last in gen_blocks(lo, hi):
        blk = _gen_block(dist, b, device, seed)
        n = (last - first) * per
        out[at:at + n] = blk[first * per:last * per]
        at += n
        del blk
    return out.view(torch.uint8)


def gen_text(n_bytes: int, seed: int = 0x5EED0006):
    """TPC-H lineitem-like text (8d config 4) from benchdata/tpch_text.c, host -> device."""
    import numpy as np
    import torch
    path = os.path.join(ROOT, "benchdata", "libbenchdata.so")
    B = ctypes.CDLL(path)
    B.benchdata_tpch_lineitem_text.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int]
    host = np.empty(n_bytes, dtype=np.uint8)
    B.benchdata_tpch_lineitem_text(host.ctypes.data, n_bytes, seed, _cores())
    return host


def _splitmix(x):
    """splitmix64 finaliser on int64 tensors (two's complement wrap-around)."""
    x = x + (-7046029254386353131)                        # 0x9E3779B97F4A7C15
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * (-7723592293110705685)  # 0x94D049BB133111EB
    return x ^ ((x >> 31) & ((1 << 33) - 1))


def gen_sorted_columns(n_parts: int, device, seed: int = 0x5EED0005, inc_bits: int = 3):
    """Config 3: one sorted uint32 column per 64 KiB partition; v[0] ~ U[0, 2^20),
    v[i] = v[i-1] + (g == 0 ? 0 : U[1, 8]), g ~ U[0, 4) (about 25 % repeats);
    partition i draws from the counter-based stream of seed + i.  inc_bits other than 3: steps of
    U[1, 2^inc_bits] (less compressible columns for measurements; not a BASELINE config)."""
    import torch
    per = CHUNK // 4
    out = torch.empty(n_parts * per, dtype=torch.int32, device=device)
    step = 8192
    j = torch.arange(per, dtype=torch.int64, device=device)
    for p0 in range(0, n_parts, step):
        p1 = min(p0 + step, n_parts)
        s = (torch.arange(p0, p1, dtype=torch.int64, device=device) + seed)[:, None]
        r = _splitmix((s << 20) + j[None, :])
        g = r & 3
        inc = torch.where(g == 0, torch.zeros_like(r), ((r >> 8) & ((1 << inc_bits) - 1)) + 1)
        inc[:, 0] = (r[:, 0] >> 16) & ((1 << 20) - 1)
        v = torch.cumsum(inc, dim=1)
        # every 1024-element sub-chunk holds at least two distinct values, so
        # no layer sees an empty array
        sub = v.view(p1 - p0, per // 1024, 1024)
        assert bool((sub[:, :, 0] != sub[:, :, -1]).all().item())
        out[p0 * per:p1 * per] = v.to(torch.int32).view(-1)
        del r, g, inc, v, sub
    return out.view(torch.uint8)


# ---- one codec over one rank's chunks ----------------------------------------

class CodecJob:
    """Buffers of one rank, allocated once like a caller of the C API would."""

    def __init__(self, hc, lib, name, opts, data):
        import torch
        self.codec = hc.batch.Codec(name, opts, lib=lib)
        self.src = hc.batch.from_device_buffer(data, CHUNK)
        n = self.src.n
        dev = data.device
        self.n = n
        self.total = int(data.numel())
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
        import torch
        assert int(self.statuses.abs().sum().item()) == 0, "decompress reported errors"
        assert bool((self.actual == self.src.sizes).all().item())
        k = self.total // 8 * 8
        a = self.out.data[:k].view(torch.int64)
        b = self.src.data[:k].view(torch.int64)
        assert bool(torch.equal(a, b)), "round trip mismatch"
        assert bool(torch.equal(self.out.data[k:self.total], self.src.data[k:self.total]))

    def compressed_bytes(self) -> int:
        return int(self.comp.sizes.sum().item())


def time_phases(job: CodecJob, steps: int):
    """Per-step HIP-event times (ms) of the two launches, on the launch stream."""
    import torch
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


def _cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_liblz4_baseline(sample, reps: int = 3):
    """System liblz4 round trip on the host cores (BASELINE.json configs[0]):
    oracle/cpu_baseline.c -- pthreads, static contiguous partition, best of
    `reps`.  Falls back to the scalar C restatement when liblz4 is absent."""
    from oracle import oracle as O
    L = O.lib()
    cores = _cores()
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


def cpu_codec_baseline(codec: str, sample, reps: int = 2):
    """Snappy / Cascaded round trip on the host cores (oracle/cpu_baseline.c:
    cpu_codec_roundtrip): libsnappy when the box has it, else a plain scalar
    Snappy encoder + the restatement's decoder; Cascaded = the C restatement."""
    from oracle import oracle as O
    L = O.lib()
    cores = _cores()
    n = sample.size // CHUNK
    tc, td, ct, used = ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t(), ctypes.c_int()
    L.cpu_codec_roundtrip.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.cpu_codec_roundtrip(1 if codec == "Snappy" else 2, sample.ctypes.data, n, CHUNK, cores, reps,
                               ctypes.byref(tc), ctypes.byref(td), ctypes.byref(ct), ctypes.byref(used))
    total = n * CHUNK
    what = ("system libsnappy (snappy_compress / snappy_uncompress)" if used.value else
            ("libsnappy unavailable: scalar libsnappy-style encoder + oracle/snappy_oracle.c decoder"
             if codec == "Snappy" else "oracle/cascaded_oracle.c (no third-party CPU equivalent)"))
    return {"value": total / (tc.value + td.value) / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "roundtrip_ok": rc == 0,
            "sample": f"{what}, {n} x 64KiB chunks of the same data, {cores} pthreads, best of {reps}; "
                      f"compress {total / tc.value / 1e9:.2f} GB/s, decompress {total / td.value / 1e9:.2f} GB/s, "
                      f"ratio {total / max(ct.value, 1):.3f}"}


def measure_row(hc, lib, codec: str, opts, data, label: dict, key: str, reps: int = 5):
    """One extra row: compress / decompress of `data` through `codec`.  `key` names the row in
    profiles/rNN_rows.json (the rocprof passes of the same workload)."""
    import torch
    job = CodecJob(hc, lib, codec, opts, data)
    job.compress(); job.decompress(); torch.cuda.synchronize()
    job.verify()
    tc, td = time_phases(job, reps)
    nb, cb = job.total, job.compressed_bytes()
    row = dict(label)
    row.update({"row": key, "chunks": job.n, "ratio": nb / max(cb, 1),
                "compress_GBps": nb / (min(tc) * 1e-3) / 1e9, "decompress_GBps": nb / (min(td) * 1e-3) / 1e9,
                "roundtrip_GBps": nb / ((min(tc) + min(td)) * 1e-3) / 1e9,
                "compress_ms": min(tc), "decompress_ms": min(td),
                "hbm_frac_compress": (nb + cb) / (min(tc) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hbm_frac_decompress": (nb + cb) / (min(td) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "roofline": {"compress": roofline_of(key, "compress", nb + cb, min(tc)),
                             "decompress": roofline_of(key, "decompress", nb + cb, min(td))}})
    if REF_LIB is not None:
        # --ref: the reference's own kernels (oracle/_ref) on the same GPU and data, outside every timed region
        rjob = CodecJob(hc, REF_LIB, codec, opts, data)
        rjob.compress(); rjob.decompress(); torch.cuda.synchronize()
        rtc, rtd = time_phases(rjob, 2)
        row["reference_gpu"] = {
            "compress_ms": min(rtc), "decompress_ms": min(rtd),
            "compress_GBps": nb / (min(rtc) * 1e-3) / 1e9, "decompress_GBps": nb / (min(rtd) * 1e-3) / 1e9,
            "compressed_sizes_identical": bool(torch.equal(rjob.comp.sizes, job.comp.sizes)),
            "speedup_compress": min(rtc) / min(tc), "speedup_decompress": min(rtd) / min(td),
        }
        del rjob
    del job
    torch.cuda.empty_cache()
    return row


REF_LIB = None   # (--ref: hc.HipcompLibrary of oracle/_ref/libhipcomp_ref.so, set in main)


def gen_mixed(n_chunks: int, device):
    """Chunk i: uniform data (no match at all) for even i, the harness's data for odd i -- a batch as a
    column store hands it over: columns that do not compress beside columns that do."""
    import torch
    h = n_chunks // 2
    a = gen_data("uniform", 0, h, device, 0x5EED0002).view(h, CHUNK)
    b = gen_data("harness", 0, h, device, 0x5EED0003).view(h, CHUNK)
    return torch.stack([a, b], dim=1).reshape(-1).contiguous()


# The LZ4 routing kernel (lz4_far.hiph:lz4_route_kernel) looks at FOUR pieces of 256 bytes of a 64 KiB chunk, at
# 1/8, 3/8, 5/8 and 7/8 of it; its first stretch at the first 128 bytes of the first and the third piece, and a
# chunk without a sign of a repeat there is looked at no further.
ROUTE_PIECES = tuple((2 * k + 1) * (CHUNK // 8) for k in range(4))   # 8192, 24576, 40960, 57344
ROUTE_PIECE_BYTES = 256


def gen_misrouted(kind: str, n_chunks: int, device):
    """Chunks that misrepresent themselves exactly where the routing kernel looks (adversarial to the sampler as it
    is: round 4's rows still corrupted the middle KiB, which the sampler no longer reads):
      "text_random_samples"  TPC-H-like text with random bytes in the four sampled pieces (+ 8 bytes in front: the
                             sample looks back that far) -- routed to the LDS shape, made for data without matches,
                             although 98.4 % of the chunk compresses;
      "random_text_samples"  random bytes with text in the four pieces -- routed to the sparse far class although
                             nothing else in the chunk matches;
      "text_random_first"    text with random bytes only where the sampler's FIRST stretch looks (2 x 128 bytes):
                             the early exit takes the chunk for data without matches.
    The bytes never depend on the routing; these rows put a number on what a wrong guess costs."""
    import torch
    text = torch.from_numpy(gen_text(n_chunks * CHUNK)).to(device).view(n_chunks, CHUNK)
    rnd = gen_data("uniform", 0, n_chunks, device, 0x5EED0007).view(torch.uint8).view(n_chunks, CHUNK)
    if kind == "text_random_first":
        spans = [(ROUTE_PIECES[0] - 8, ROUTE_PIECES[0] + 128 + 4), (ROUTE_PIECES[2] - 8, ROUTE_PIECES[2] + 128 + 4)]
    else:
        spans = [(p - 8, p + ROUTE_PIECE_BYTES + 4) for p in ROUTE_PIECES]
    if kind in ("text_random_samples", "text_random_first"):
        out, other = text.clone(), rnd
    elif kind == "random_text_samples":
        out, other = rnd.clone(), text
    else:
        raise ValueError(kind)
    for lo, hi in spans:
        out[:, lo:hi] = other[:, lo:hi]
    return out.reshape(-1).contiguous()


MISROUTED_KINDS = ("text_random_samples", "random_text_samples", "text_random_first")


def measure_hlif(hc, data, codec: str = "LZ4", reps: int = 2, lib_path=None):
    """the high-level managers (container = header + offsets / sizes / checksums + chunks,
    include/hipcomp/hlif.h) over one buffer, beside the batched calls they wrap."""
    import torch
    from ctypes import c_int, c_size_t, c_void_p
    L = ctypes.CDLL(lib_path or hc.default_library().path)
    h = c_void_p()
    if codec == "LZ4":
        assert L.hipcompHlifLZ4ManagerCreate(c_size_t(CHUNK), c_int(0), None, ctypes.byref(h)) == 0
    elif codec == "Snappy":
        assert L.hipcompHlifSnappyManagerCreate(c_size_t(CHUNK), None, ctypes.byref(h)) == 0
    else:
        class _Opts(ctypes.Structure):
            _fields_ = [("chunk_size", c_size_t), ("type", c_int), ("num_RLEs", c_int), ("num_deltas", c_int), ("use_bp", c_int)]
        L.hipcompHlifCascadedManagerCreate.argtypes = [_Opts, c_void_p, c_void_p]
        assert L.hipcompHlifCascadedManagerCreate(_Opts(CHUNK, int(hc.hipcompType.UINT), 2, 1, 1), None, ctypes.byref(h)) == 0
    n = int(data.numel())
    mx, nc = c_size_t(0), c_size_t(0)
    assert L.hipcompHlifConfigureCompression(h, c_size_t(n), ctypes.byref(mx), ctypes.byref(nc)) == 0
    dst = torch.empty(mx.value + 8, dtype=torch.uint8, device=data.device)
    back = torch.empty(n, dtype=torch.uint8, device=data.device)

    def comp():
        assert L.hipcompHlifCompress(h, c_void_p(data.data_ptr()), c_size_t(n), c_void_p(dst.data_ptr())) == 0

    def dec():
        assert L.hipcompHlifDecompress(h, c_void_p(dst.data_ptr()), c_void_p(back.data_ptr())) == 0
    comp(); dec(); torch.cuda.synchronize()
    st = c_int(-1)
    assert L.hipcompHlifGetLastStatus(h, ctypes.byref(st)) == 0 and st.value == 0
    assert bool(torch.equal(back, data)), "HLIF round trip mismatch"
    size = c_size_t(0)
    assert L.hipcompHlifGetCompressedSize(h, c_void_p(dst.data_ptr()), ctypes.byref(size)) == 0
    tcs, tds = [], []
    for _ in range(reps):       # (host wall time around a stream sync: decompress reads the header on the host)
        torch.cuda.synchronize(); t0 = time.perf_counter(); comp(); torch.cuda.synchronize(); tcs.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); dec(); torch.cuda.synchronize(); tds.append(time.perf_counter() - t0)
    L.hipcompHlifManagerDestroy(h)
    what = {"LZ4": "uniform/char", "Snappy": "text", "Cascaded": "sorted"}[codec]
    return {"codec": f"{codec} high-level manager (HLIF)", "row": f"hlif/{codec.lower()}/{what}", "chunks": nc.value,
            "container_bytes": size.value, "ratio": n / max(size.value, 1),
            "compress_GBps": n / min(tcs) / 1e9, "decompress_GBps": n / min(tds) / 1e9,
            "compress_ms": min(tcs) * 1e3, "decompress_ms": min(tds) * 1e3,
            "note": "host wall time incl. the stream synchronisation; the batched encoders place every chunk in the "
                    "container themselves when its size is known (completion order, like the reference's managers)"}


def _code_only(text: bytes) -> bytes:
    """C / C++ source without comments and with white space collapsed."""
    import re
    t = text.decode("utf-8", "replace")
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r"\s+", " ", t).encode()


def kernel_source_id(read=None) -> str:
    """Identifies the kernel build a rocprof pass belongs to: sha256 over the device sources, comments
    and white space left out (a reworded comment does not orphan the committed profiles).
    `read(path) -> bytes`: another way to get at the files (scripts/rekey_rows.py reads a commit)."""
    import glob
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hipcomp-core_amd", "csrc")
    if read is None:
        def read(path):
            with open(path, "rb") as f:
                return f.read()
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hiph")) + glob.glob(os.path.join(d, "*.hpp"))):
        h.update(os.path.basename(path).encode())
        h.update(_code_only(read(path)))
    return h.hexdigest()[:16]


def device_asm_id():
    """Identifies the DEVICE code of the build at hand: sha256 over the assembly the kernel objects were made from
    (csrc/build/*.gfx950.s, kept by the Makefile).  A change to host code or comments in a kernel source file
    changes kernel_source_id() but not this; None when the build directory is not there."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "hipcomp-core_amd", "csrc", "build", "*.gfx950.s")))
    if not files:
        return None
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


PROFILED_ROWS = None


def profiled(label: str, phase: str):
    """What the committed rocprofv3 passes say about one bench row's kernel (profiles/rNN_rows.json,
    written by scripts/collect_profiles.sh + scripts/profile_rows.py): average duration from
    --kernel-trace --stats, HBM-side bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes
    (FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE; not doubled for
    the kernels whose reads are narrow gathers -- scripts/profile_rows.py says which and why).  None when
    there is no pass for this row OR the passes were taken on another build of the kernels (the file
    records the sha256 of the kernel source files it was measured on and of the device assembly made from them)."""
    global PROFILED_ROWS
    if PROFILED_ROWS is None:
        PROFILED_ROWS = {}
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_rows.json")), reverse=True):  # (the newest round first)
            with open(path) as f:
                table = json.load(f)
            # (the same sources, or -- host code or comments of a kernel file changed since -- the same device assembly)
            if table.get("kernel_source_sha16") == kernel_source_id() or (
                    table.get("device_asm_sha16") and table.get("device_asm_sha16") == device_asm_id()):
                PROFILED_ROWS = table.get("rows", {})
                break
    return PROFILED_ROWS.get(label, {}).get(phase)


def roofline_of(label: str, phase: str, algo_bytes: int, ms: float):
    """The roofline object of one row and phase: algorithmic bytes (N + C) over the HIP-event
    time of the launch measured here, beside what the committed rocprof passes of the same row hold."""
    ach = algo_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "algorithmic_bytes_per_launch": algo_bytes, "traffic": None}
    p = profiled(label, phase)
    if p:
        r.update({"kernel": p.get("kernel"), "traffic": p.get("traffic_bytes_per_launch"),
                  "rocprof_avg_ms": p.get("avg_ms"),
                  "traffic_over_algorithmic": (p["traffic_bytes_per_launch"] / algo_bytes) if p.get("traffic_bytes_per_launch") else None})
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=100000, help="chunks per GPU (weak scaling, the headline)")
    ap.add_argument("--total-chunks", type=int, default=CONFIG5_TOTAL,
                    help="chunks in all for the strong-scaling row (BASELINE config 5: 10 GiB)")
    ap.add_argument("--config5", dest="config5", action="store_true", default=None,
                    help="also run the strong-scaling row at N = 1 (default: only when N > 1)")
    ap.add_argument("--no-config5", dest="config5", action="store_false")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "harness", "runs"])
    ap.add_argument("--dtype", default="char", choices=["char", "int"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ref", action="store_true", help="also time the reference build (oracle/_ref) on the same buffers")
    ap.add_argument("--cpu-sample-chunks", type=int, default=16384)
    ap.add_argument("--backend", default="gloo", help="torch.distributed backend for the barrier and the two scalar "
                    "reductions; the data path has no collective, so the default is gloo (no RCCL)")
    ap.add_argument("--no-variants", dest="variants", action="store_false",
                    help="skip the extra rows (other distributions / data types / codecs)")
    ap.add_argument("--variant-chunks", type=int, default=100000)
    ap.add_argument("--text-chunks", type=int, default=65536, help="chunks of TPC-H-like text for the config-4 rows (4 GiB)")
    ap.add_argument("--dry-run", action="store_true",
                    help="process plumbing only (spawn, rendezvous, barrier, slice arithmetic, reductions) with no "
                         "GPU and no codec call; prints a line whose value is null -- for the CPU tests")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing has touched the GPU in this process.
        sys.exit(self_spawn(sys.argv[1:], args.gpus))

    # Exactly ONE line goes to stdout, the JSON line: everything else this process
    # or a library it loads writes there (gloo announces its ranks on stdout) is sent
    # to stderr from here on.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
        lo, hi = shard_slice(args.chunks * world, world, rank)
        lo5, hi5 = shard_slice(args.total_chunks, world, rank)
        v, w = aggregate_throughput(1.0 + rank, (hi - lo) * CHUNK, args.steps, dist if world > 1 else None, "cpu")
        v5, _ = aggregate_throughput(1.0, (hi5 - lo5) * CHUNK, args.steps, dist if world > 1 else None, "cpu")
        if rank == 0:
            emit({"dry_run": True, "value": None, "n_gpus": world, "weak_bytes_per_step_GB": v * w / args.steps,
                  "strong_bytes_per_step_GB": v5 * 1.0 / args.steps, "slowest_rank_wall_s": w})
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    local = local % max(torch.cuda.device_count(), 1)  # (rehearsal: several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1
    dist = None
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    hc = importlib.import_module("hipcomp-core_amd")
    lib = hc.default_library()
    if args.ref:
        from oracle import oracle as O
        if os.path.exists(O.REF_LIB_PATH):
            global REF_LIB
            REF_LIB = hc.HipcompLibrary(O.REF_LIB_PATH)
    dtype = hc.hipcompType.CHAR if args.dtype == "char" else hc.hipcompType.INT
    opts = hc.LZ4Opts(dtype)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def run_job(lo: int, hi: int, seed: int):
        """Warm up, then time exactly args.steps steps of the rank's slice."""
        data = gen_data(args.dist, lo, hi, dev, seed)
        job = CodecJob(hc, lib, "LZ4", opts, data)
        for _ in range(max(args.warmup, 1)):
            job.compress()
            job.decompress()
        torch.cuda.synchronize()
        job.verify()
        barrier()
        t0 = time.perf_counter()
        tc, td = time_phases(job, args.steps)
        barrier()
        wall_local = time.perf_counter() - t0
        value, wall = aggregate_throughput(wall_local, job.total, args.steps, dist, dev)
        return data, job, tc, td, value, wall

    # ---- headline: weak scaling, rank r owns chunks [r*chunks, (r+1)*chunks) of one list
    lo, hi = shard_slice(args.chunks * world, world, rank)
    data, job, tc, td, value, wall = run_job(lo, hi, 0x5EED0002)
    n_bytes, c_bytes = job.total, job.compressed_bytes()
    ms_step = wall / args.steps * 1e3
    res = None
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
                "parallelism": f"{world} contiguous slices of one seeded chunk list, one process per GPU, "
                               f"no data-path collective (barrier/reductions over {args.backend if dist_on else 'nothing'})",
            },
            "ratio": n_bytes / max(c_bytes, 1),
            "compress_GBps": n_bytes / (tc_avg * 1e-3) / 1e9,
            "decompress_GBps": n_bytes / (td_avg * 1e-3) / 1e9,
            "compress_ms": tc_avg, "decompress_ms": td_avg,
            "roofline": dict(roofline_of(f"lz4/{args.dist}/{args.dtype}/{args.chunks}", "compress", algo, tc_avg),
                             kernel="lz4_compress_kernel_pair" if args.dist == "uniform" else "lz4_compress_kernel_far",
                             decompress_achieved=algo / (td_avg * 1e-3) / 1e9,
                             decompress_frac=algo / (td_avg * 1e-3) / 1e9 / HBM_PEAK_GBS),
        }
        if args.ref:
            from oracle import oracle as O
            if os.path.exists(O.REF_LIB_PATH):
                rjob = CodecJob(hc, hc.HipcompLibrary(O.REF_LIB_PATH), "LZ4", opts, data)
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
        if not args.no_cpu and world == 1:  # reported at N=1 only
            k = min(args.cpu_sample_chunks, job.n)
            res["cpu_baseline"] = cpu_liblz4_baseline(data[: k * CHUNK].cpu().numpy())
    del job, data
    torch.cuda.empty_cache()

    # ---- BASELINE config 5: ONE list of --total-chunks, strong scaling
    want5 = args.config5 if args.config5 is not None else world > 1
    if want5:
        lo, hi = shard_slice(args.total_chunks, world, rank)
        data, job, tc5, td5, value5, wall5 = run_job(lo, hi, 0x5EED0002)
        if rank == 0:
            res["config5_strong"] = {
                "scaling": "strong", "total_chunks": args.total_chunks, "total_bytes": args.total_chunks * CHUNK,
                "chunks_rank0": hi - lo, "value": value5, "unit": "GB/s", "ms_per_step": wall5 / args.steps * 1e3,
                "rank0_compress_ms": sum(tc5) / len(tc5), "rank0_decompress_ms": sum(td5) / len(td5),
            }
        del job, data
        torch.cuda.empty_cache()

    # ---- every other BASELINE config, N = 1 only
    if rank == 0 and args.variants and world == 1:
        rows = []
        vc = args.variant_chunks
        seeds = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
        for i, (dn, tn) in enumerate((("uniform", "int"), ("harness", "char"), ("harness", "int"),
                                      ("runs", "char"), ("runs", "int"))):
            d = gen_data(dn, 0, vc, dev, seeds[dn])
            t = hc.hipcompType.CHAR if tn == "char" else hc.hipcompType.INT
            rows.append(measure_row(hc, lib, "LZ4", hc.LZ4Opts(t), d,
                                    {"codec": "LZ4", "distribution": dn, "data_type": tn.upper()}, f"lz4/{dn}/{tn}/{vc}"))
            del d
        # the headline next to its five siblings: the geometric mean of the round trips of the three
        # distributions x {CHAR, INT} (the headline alone is the friendliest of the six)
        six = [res["value"]] + [r["roundtrip_GBps"] for r in rows]
        g = 1.0
        for v in six:
            g *= v
        res["geomean_roundtrip_GBps"] = g ** (1.0 / len(six))
        res["geomean_of"] = "LZ4 round trip GB/s of {uniform, harness, runs} x {CHAR, INT}, %d chunks each" % vc
        # a batch that mixes chunks without matches and chunks that compress (the routing kernel sends every
        # chunk to its shape): against the time-weighted sum of its halves measured alone
        d = gen_mixed(vc, dev)
        mrow = measure_row(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d,
                           {"codec": "LZ4", "distribution": "mixed: chunk i uniform for even i, harness for odd i", "data_type": "CHAR"},
                           f"lz4/mixed/char/{vc}")
        halves = (res["compress_ms"] * vc / args.chunks + next(r for r in rows if r["row"] == f"lz4/harness/char/{vc}")["compress_ms"]) / 2
        mrow["compress_ms_time_weighted_sum_of_halves"] = halves
        mrow["compress_over_halves"] = mrow["compress_ms"] / halves
        rows.append(mrow)
        del d
        # batches far smaller than the chip holds waves (the reference harness's own sizes): 1000 chunks
        for dn in ("harness", "text"):
            d = (gen_data(dn, 0, 1000, dev, seeds[dn]) if dn != "text" else torch.from_numpy(gen_text(1000 * CHUNK)).to(dev))
            rows.append(measure_row(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d,
                                    {"codec": "LZ4", "distribution": dn if dn != "text" else "tpch_lineitem_text",
                                     "data_type": "CHAR", "note": "a small batch"}, f"lz4/{dn}/char/1000", reps=5))
            del d
        # what a wrong guess of the routing kernel costs (it looks at 1 KiB from the middle of a chunk)
        for kind in MISROUTED_KINDS:
            d = gen_misrouted(kind, 16384, dev)
            rows.append(measure_row(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d,
                                    {"codec": "LZ4", "distribution": "misrouted: " + kind, "data_type": "CHAR",
                                     "note": "exactly the bytes the routing kernel samples (4 x 256 of 65 536; "
                                             "'first': 2 x 128) misrepresent the chunk: what a wrong guess costs"},
                                    f"lz4/misrouted_{kind}/char/16384", reps=3))
            del d
        tchunks = args.text_chunks
        text_host = gen_text(tchunks * CHUNK)
        text = torch.from_numpy(text_host).to(dev)
        rows.append(measure_row(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), text,
                                {"codec": "LZ4", "distribution": "tpch_lineitem_text", "data_type": "CHAR"}, f"lz4/text/char/{tchunks}"))
        srow = measure_row(hc, lib, "Snappy", hc.SnappyOpts(0), text,
                           {"codec": "Snappy", "config": f"BASELINE configs[3]: TPC-H lineitem-like text, {tchunks} x 64 KiB chunks"},
                           f"snappy/text/{tchunks}")
        if not args.no_cpu:
            srow["cpu_baseline"] = cpu_codec_baseline("Snappy", text_host[: 8192 * CHUNK])
        rows.append(srow)
        rows.append(measure_hlif(hc, text.view(torch.uint8), "Snappy"))
        del text, text_host
        cols = gen_sorted_columns(vc, dev)
        crow = measure_row(hc, lib, "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), cols,
                           {"codec": "Cascaded", "config": "BASELINE configs[2]: sorted uint32 columns, opts {4096, UINT, 2, 1, 1}"},
                           f"cascaded/sorted/{vc}")
        if not args.no_cpu:
            crow["cpu_baseline"] = cpu_codec_baseline("Cascaded", cols[: 8192 * CHUNK].cpu().numpy())
        rows.append(crow)
        rows.append(measure_hlif(hc, cols.view(torch.uint8), "Cascaded"))
        # the option selector (include/hipcomp/cascaded_select.h, an API of this library's own): what it picks
        # for the config-3 columns and for a column that should not be cascaded at all, and the ratio that buys
        for cname, cdata in (("sorted", cols), ("uniform", gen_data("uniform", 0, 16384, dev, seeds["uniform"]).view(torch.uint8))):
            job = CodecJob(hc, lib, "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), cdata)
            temp = torch.empty(lib.cascaded_select_temp_size(), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opts, est = lib.cascaded_select_opts(job.src.ptrs.data_ptr(), job.src.sizes.data_ptr(), job.n, hc.hipcompType.UINT,
                                                 temp.data_ptr(), temp.numel(), torch.cuda.current_stream().cuda_stream)
            select_ms = (time.perf_counter() - t0) * 1e3
            del job, temp
            srow2 = measure_row(hc, lib, "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, opts.num_RLEs, opts.num_deltas, opts.use_bp),
                                cdata, {"codec": "Cascaded", "config": f"options picked by hipcompBatchedCascadedSelectOpts for the {cname} columns",
                                        "selected_opts": {"num_RLEs": opts.num_RLEs, "num_deltas": opts.num_deltas, "use_bp": opts.use_bp},
                                        "estimated_ratio": est, "select_ms": select_ms},
                                f"cascaded/{cname}/selected", reps=3)
            rows.append(srow2)
        del cols, cdata
        d = gen_data("uniform", 0, vc, dev, seeds["uniform"]).view(torch.uint8)
        rows.append(measure_hlif(hc, d))
        del d
        res["extra_keys"] = rows
    if rank == 0:
        emit(res)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
