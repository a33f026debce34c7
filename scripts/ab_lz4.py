"""A/B timing of LZ4 builds in ONE process, interleaved rounds (guide rule 24).

usage: ab_lz4.py [--chunks N] [--dist uniform|harness|runs|text] [--dtype char|int] lib1.so lib2.so ...
Reports min/median compress and decompress ms per library and whether the
compressed bytes equal those of the first library.
"""
import argparse, importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--dtype", default="char")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
if a.dist == "text":
    import datagen, numpy as np
    one = np.frombuffer(datagen.text_like(1, 1 << 22), dtype=np.uint8)
    reps = (a.chunks * bench.CHUNK + one.size - 1) // one.size
    data = torch.from_numpy(np.tile(one, reps)[: a.chunks * bench.CHUNK].copy()).to(dev)
else:
    data = bench.gen_data(a.dist, a.chunks, dev, seed=0x5EED0002)
dtype = hc.hipcompType.CHAR if a.dtype == "char" else hc.hipcompType.INT
jobs = []
for path in a.libs:
    p = path if os.path.isabs(path) else os.path.join(ROOT, path)
    job = bench.Lz4Job(hc, hc.HipcompLibrary(p, codecs=("LZ4",)), data, dtype)
    job.comp.data.zero_()
    job.compress(); job.decompress(); torch.cuda.synchronize()
    jobs.append(job)
tc = [[] for _ in jobs]; td = [[] for _ in jobs]
for r in range(a.rounds):
    for i, job in enumerate(jobs):
        c, d = bench.time_phases(job, 1)
        tc[i] += c; td[i] += d
n_bytes = jobs[0].n * bench.CHUNK
for i, job in enumerate(jobs):
    same = bool(torch.equal(job.comp.sizes, jobs[0].comp.sizes)) and bool(torch.equal(job.comp.data, jobs[0].comp.data))
    try:
        job.verify(); rt = "ok"
    except AssertionError as e:
        rt = "ROUNDTRIP-FAIL"
    cb = int(job.comp.sizes.sum().item())
    print(f"{os.path.basename(a.libs[i]):28s} comp min {min(tc[i]):9.3f} med {statistics.median(tc[i]):9.3f} ms "
          f"({n_bytes / min(tc[i]) / 1e6:7.1f} GB/s) | decomp min {min(td[i]):8.3f} ms ({n_bytes / min(td[i]) / 1e6:7.1f} GB/s) "
          f"| ratio {n_bytes / max(cb,1):6.3f} | same_as_first={same} roundtrip={rt}", flush=True)
