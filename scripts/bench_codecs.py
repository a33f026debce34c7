"""Snappy and Cascaded throughput (BASELINE.json configs[2], configs[3]) for this
library and, when present, the reference build -- same GPU, same buffers.

usage: bench_codecs.py [--chunks N] [--codec snappy|cascaded|both] [--rounds K]
Prints one JSON line per (codec, library).
"""
import argparse, importlib, json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import datagen

CHUNK = 65536
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--codec", default="both")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--no-ref", action="store_true")
ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline line")
ap.add_argument("--cpu-sample-chunks", type=int, default=1024)
ap.add_argument("--extra-lib", action="append", default=[], help="name=path of another build of this library to time next to it")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
from oracle import oracle as O
dev = torch.device("cuda:0")
libs = [("ours", hc.default_library())]
for spec in a.extra_lib:
    nm, path = spec.split("=", 1)
    libs.append((nm, hc.HipcompLibrary(path if os.path.isabs(path) else os.path.join(ROOT, path))))
if not a.no_ref and os.path.exists(O.REF_LIB_PATH):
    libs.append(("reference", hc.HipcompLibrary(O.REF_LIB_PATH)))


def tile(host: np.ndarray, n_chunks: int) -> torch.Tensor:
    base = torch.from_numpy(host).to(dev)
    reps = (n_chunks * CHUNK + base.numel() - 1) // base.numel()
    return base.repeat(reps)[: n_chunks * CHUNK].contiguous()


def run(codec_name, opts, data):
    src = hc.batch.from_device_buffer(data, CHUNK)
    n = src.n
    results = []
    first = None
    for name, lib in libs:
        codec = hc.batch.Codec(codec_name, opts, lib=lib)
        comp = hc.batch.alloc_batch(n, codec.max_output_chunk_size(CHUNK), dev, fill=0)
        out = hc.batch.alloc_batch(n, CHUNK, dev)
        caps = torch.full((n,), CHUNK, dtype=torch.int64, device=dev)
        actual = torch.zeros(n, dtype=torch.int64, device=dev)
        stat = torch.zeros(n, dtype=torch.int32, device=dev)
        tc, td = [], []
        for r in range(a.rounds + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            assert codec.compress_async(src, CHUNK, None, comp) == 0
            e[1].record()
            assert codec.decompress_async(comp, caps, actual, None, out, stat) == 0
            e[2].record()
            torch.cuda.synchronize()
            if r:
                tc.append(e[0].elapsed_time(e[1])); td.append(e[1].elapsed_time(e[2]))
        ok = int(stat.abs().sum().item()) == 0 and bool(torch.equal(out.data[: n * CHUNK].view(torch.int64), data.view(torch.int64)))
        cb = int(comp.sizes.sum().item())
        same = None
        if first is None:
            first = comp
        else:
            same = bool(torch.equal(comp.sizes, first.sizes))
            if codec_name == "Snappy":
                same = same and bool(torch.equal(comp.data, first.data))
        nb = n * CHUNK
        results.append({"codec": codec_name, "lib": name, "chunks": n, "ratio": nb / max(cb, 1),
                        "compress_ms": min(tc), "decompress_ms": min(td),
                        "compress_GBps": nb / min(tc) / 1e6, "decompress_GBps": nb / min(td) / 1e6,
                        "roundtrip_GBps": nb / (min(tc) + min(td)) / 1e6,
                        "hbm_frac_compress": (nb + cb) / min(tc) / 1e6 / 8000, "hbm_frac_decompress": (nb + cb) / min(td) / 1e6 / 8000,
                        "roundtrip_ok": ok, "same_output_as_ours": same})
        print(json.dumps(results[-1]), flush=True)
        del comp, out
    return results


def cpu_baseline(codec_name, host: np.ndarray):
    """CPU port (oracle/, scalar C) of the same codec on a bounded sample of the same data, one
    Python thread per host core (ctypes releases the GIL).  A reported baseline, not a target:
    SURVEY 8(d) asks for libsnappy when present -- it is not in this image -- and else for the
    own CPU codec."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    n = min(a.cpu_sample_chunks, host.size // CHUNK)
    chunks = [host[i * CHUNK:(i + 1) * CHUNK].tobytes() for i in range(n)]
    if codec_name == "Snappy":
        enc = O.snappy_compress
        dec = lambda z: O.snappy_decompress(z, CHUNK)
    else:
        enc = lambda c: O.cascaded_compress(c, 5, 2, 1, 1)[0]
        dec = lambda z: O.cascaded_decompress(z, CHUNK)
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter(); comp = list(ex.map(enc, chunks)); t1 = time.perf_counter()
        outs = list(ex.map(dec, comp)); t2 = time.perf_counter()
    ok = all(o[0] == 0 and o[1] == c for o, c in zip(outs, chunks))
    nb = n * CHUNK
    print(json.dumps({"codec": codec_name, "lib": "cpu_baseline", "kind": "port", "cores": cores, "chunks": n,
                      "compress_GBps": nb / (t1 - t0) / 1e9, "decompress_GBps": nb / (t2 - t1) / 1e9,
                      "roundtrip_GBps": nb / (t2 - t0) / 1e9, "roundtrip_ok": ok,
                      "sample": f"oracle/ C port (scalar restatement of the reference codec), {n} x 64 KiB chunks of the same data, "
                                f"{cores} threads; libsnappy is not installed in this image"}), flush=True)


if a.codec in ("snappy", "both"):
    text = np.frombuffer(datagen.tpch_lineitem_text(0x5EED0006, 1 << 24), dtype=np.uint8).copy()
    run("Snappy", hc.SnappyOpts(0), tile(text, a.chunks))
    if not a.no_cpu:
        cpu_baseline("Snappy", text)
if a.codec in ("cascaded", "both"):
    # sorted uint32 columns, ~25% repeats, increments 1..8, one column per 64 KiB partition
    base = np.concatenate([datagen.sorted_column(0x5EED0005 + i, CHUNK // 4) for i in range(256)]).view(np.uint8)
    run("Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), tile(base, a.chunks))
    if not a.no_cpu:
        cpu_baseline("Cascaded", base)
