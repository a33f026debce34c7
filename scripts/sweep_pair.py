"""The pair shape beside the lone-wave mix shape over chunk sizes and batch sizes (knobs build, HIPCOMP_LZ4_PAIR),
uniform data: sweep_pair.py [--dtype char,int] [--modes 0,1]   -- bytes compared with the first mode's."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="char")
ap.add_argument("--modes", default="0,1")
ap.add_argument("--cases", default="65536:100000,65536:20000,65536:3000,65536:1000,32768:40000,16384:80000,8192:160000,90112:14000")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
lib = hc.knobs_library()
whole = bench.gen_data("uniform", 0, 100000, dev, 0x5EED0002)
for case in a.cases.split(","):
    size, n = (int(x) for x in case.split(":"))
    bench.CHUNK = size
    data = whole[: size * n]
    for dt in a.dtype.split(","):
        t = hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT
        job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(t), data)
        first = None
        line = f"{size:6d} B x {n:6d} as {dt}:"
        for mode in a.modes.split(","):
            os.environ["HIPCOMP_LZ4_PAIR"] = mode
            job.comp.data.zero_(); job.comp.sizes.zero_()
            job.compress(); torch.cuda.synchronize()
            snap = (job.comp.sizes.clone(), job.comp.data.clone())
            if first is None:
                first = snap
                job.decompress(); torch.cuda.synchronize(); job.verify()
            same = bool(torch.equal(snap[0], first[0])) and bool(torch.equal(snap[1], first[1]))
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); job.compress(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            line += f"  mode {mode}: {min(ts):8.3f} ms {job.total / min(ts) / 1e6:7.1f} GB/s same={same}"
        print(line, flush=True)
        del job
    torch.cuda.empty_cache()
