"""Where the routing kernel sends the chunks of the bench distributions: route_counts.py [--chunks N]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="uniform,harness,runs,text")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
os.environ["HIPCOMP_LZ4_SHAPE"] = "auto"
for dist in a.dist.split(","):
    if dist == "text":
        data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
    else:
        data = bench.gen_data(dist, 0, a.chunks, dev, {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}[dist])
    for dt in ("char", "int"):
        t = hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT
        job = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(t), data)
        job.compress(); torch.cuda.synchronize()
        h = job.temp[:64].view(torch.int32).cpu().tolist()
        print(f"{dist:8s} {dt:4s} n={job.n}: lists mix/dense/sparse/wide {h[4:8]}  tickets {h[0:4]}  samples repeats/looked/near {h[8:11]}"
              f"  -> {h[8] / max(h[9], 1):.3f} {h[10] / max(h[9], 1):.3f}", flush=True)
        del job
    del data
