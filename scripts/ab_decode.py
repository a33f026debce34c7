"""A/B of decoder builds in ONE process on the same compressed batches:
   ab_decode.py [--chunks N] [--dist harness,text,runs,uniform] [--codec lz4|snappy] VARIANT [VARIANT ...]
VARIANT = "" (the product) or the name of lib/libhipcomp_<name>.so; interleaved rounds, best of each."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="harness,text,runs")
ap.add_argument("--codec", default="lz4")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
SEEDS = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
libs = {v: (hc.HipcompLibrary(os.path.join(ROOT, "hipcomp-core_amd", "lib", f"libhipcomp_{v}.so")) if v and v != "product" else hc.default_library())
        for v in a.variants}
for dist in a.dist.split(","):
    if dist == "text":
        data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
    else:
        data = bench.gen_data(dist, 0, a.chunks, dev, SEEDS[dist])
    jobs = {}
    for v, lib in libs.items():
        if a.codec == "lz4":
            jobs[v] = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), data)
        else:
            jobs[v] = bench.CodecJob(hc, lib, "Snappy", hc.SnappyOpts(0), data)
        jobs[v].compress(); jobs[v].decompress(); torch.cuda.synchronize()
        jobs[v].verify()
    best = {v: (1e9, 1e9) for v in libs}
    for r in range(a.rounds):
        for v, job in jobs.items():
            tc, td = bench.time_phases(job, 2)
            best[v] = (min(best[v][0], min(tc)), min(best[v][1], min(td)))
    nb = next(iter(jobs.values())).total
    print(f"{a.codec}/{dist} n={a.chunks}: " + " | ".join(
        f"{v or 'product'}: compress {nb / best[v][0] / 1e6:7.1f} decompress {nb / best[v][1] / 1e6:7.1f} GB/s" for v in libs), flush=True)
    del jobs, data
    torch.cuda.empty_cache()
