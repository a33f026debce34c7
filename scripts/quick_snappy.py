"""Quick Snappy compress/decompress timing of the built library on one GPU (TPC-H-like text, 64 KiB chunks):
   quick_snappy.py [--chunks N] [--check]
--check compares every chunk's compressed bytes with the reference build (oracle/_ref) when present."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=16384)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--check", action="store_true")
ap.add_argument("--lib", default=None, help="another build of the library (measurement variants)")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
lib = hc.HipcompLibrary(os.path.join(ROOT, a.lib)) if a.lib else hc.default_library()
job = bench.CodecJob(hc, lib, "Snappy", hc.SnappyOpts(0), data)
job.compress(); job.decompress(); torch.cuda.synchronize()
job.verify()
tc, td = bench.time_phases(job, a.reps)
nb, cb = job.total, job.compressed_bytes()
line = f"snappy text n={job.n}: compress {min(tc):8.3f} ms {nb/min(tc)/1e6:8.1f} GB/s | decompress {min(td):8.3f} ms {nb/min(td)/1e6:8.1f} GB/s | ratio {nb/cb:.3f}"
if a.check:
    from oracle import oracle as O
    if os.path.exists(O.REF_LIB_PATH):
        rjob = bench.CodecJob(hc, hc.HipcompLibrary(O.REF_LIB_PATH), "Snappy", hc.SnappyOpts(0), data)
        rjob.compress(); torch.cuda.synchronize()
        same = bool(torch.equal(rjob.comp.sizes, job.comp.sizes))
        if same:
            stride = job.comp.stride
            idx = torch.arange(stride, device=dev)[None, :] < job.comp.sizes[:, None]
            A = job.comp.data[: job.n * stride].view(job.n, stride)
            B = rjob.comp.data[: job.n * stride].view(job.n, stride)
            same = bool(((A == B) | ~idx).all().item())
        line += f" | same_as_reference={same}"
print(line)
