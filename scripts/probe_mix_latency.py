"""Timing probe for the headline kernel: is it waiting for HBM?  The same batch with every input pointer
aimed at chunk 0 (64 KiB that stay in L2): what the kernel computes per chunk is the same, what it
waits for is not.  probe_mix_latency.py [--chunks N]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=100000)
ap.add_argument("--codec", default="lz4", help="lz4 | snappy")
ap.add_argument("--lib", default=None, help="another build of the library, e.g. hipcomp-core_amd/lib/libhipcomp_smalltab.so (timing-only: no verify)")
ap.add_argument("--dist", default="uniform", help="uniform | harness | runs | text (the far kernels: candidates come from the same few chunks too)")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
LIB = hc.HipcompLibrary(os.path.join(ROOT, a.lib)) if a.lib else hc.default_library()
data = (torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev) if a.dist == "text"
        else bench.gen_data(a.dist, 0, a.chunks, dev, {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}[a.dist]))
for dt, name in ((hc.hipcompType.CHAR, "char"), (hc.hipcompType.INT, "int")) if a.codec == "lz4" else ((0, "snappy"),):
    job = (bench.CodecJob(hc, LIB, "LZ4", hc.LZ4Opts(dt), data) if a.codec == "lz4"
           else bench.CodecJob(hc, LIB, "Snappy", hc.SnappyOpts(0), data))
    job.compress(); torch.cuda.synchronize()
    tc, _ = bench.time_phases(job, 5)
    base = min(tc)
    for k in (1, 256, 4096):
        job.src.ptrs[:] = job.src.ptrs[:k].repeat((a.chunks + k - 1) // k)[: a.chunks]   # inputs: k distinct chunks only
        job.compress(); torch.cuda.synchronize()
        tc2, _ = bench.time_phases(job, 5)
        print(f"{a.dist} {name} n={a.chunks}: compress {base:.3f} ms ({job.total / base / 1e6:.1f} GB/s); all inputs from {k} chunk(s) ({k * 64} KiB): {min(tc2):.3f} ms ({job.total / min(tc2) / 1e6:.1f} GB/s)", flush=True)
    del job
