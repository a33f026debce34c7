"""Quick Cascaded compress/decompress timing of the built library on one GPU (sorted uint32 columns, 64 KiB partitions):
   quick_cascaded.py [--parts N] [--libs ,occ5,...]   (--libs: variants lib/libhipcomp_<name>.so, "" = the product)"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=100000)
ap.add_argument("--libs", default="")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--inc-bits", type=int, default=3, help="steps of U[1, 2^n]: 3 = BASELINE config 3, more = less compressible")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
cols = bench.gen_sorted_columns(a.parts, dev, inc_bits=a.inc_bits)
for name in a.libs.split(","):
    cb = 4096
    lib = hc.HipcompLibrary(os.path.join(ROOT, "hipcomp-core_amd", "lib", f"libhipcomp_{name}.so")) if name else hc.default_library()
    job = bench.CodecJob(hc, lib, "Cascaded", hc.CascadedOpts(cb, hc.hipcompType.UINT, 2, 1, 1), cols)
    job.compress(); job.decompress(); torch.cuda.synchronize()
    job.verify()
    tc, td = bench.time_phases(job, a.reps)
    nb, cbytes = job.total, job.compressed_bytes()
    print(f"cascaded {name or 'product':8s} n={job.n}: compress {min(tc):8.3f} ms {nb/min(tc)/1e6:8.1f} GB/s | decompress {min(td):8.3f} ms {nb/min(td)/1e6:8.1f} GB/s | ratio {nb/cbytes:.3f}")
    del job
