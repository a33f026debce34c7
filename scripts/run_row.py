"""One bench row's workload, alone (for rocprofv3 passes that must see only that row's kernels):
   run_row.py lz4/harness/char/100000 | lz4/mixed/char/100000 | lz4/text/char/65536 | snappy/text/65536 | cascaded/sorted/100000
   [--reps N]   (row keys as bench.py writes them into extra_keys[].row)"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("row")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
parts = a.row.split("/")
codec, n = parts[0], int(parts[-1])
seeds = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
if codec == "lz4":
    dist, dt = parts[1], parts[2]
    if dist == "text":
        data = torch.from_numpy(bench.gen_text(n * bench.CHUNK)).to(dev)
    elif dist == "mixed":
        data = bench.gen_mixed(n, dev)
    elif dist.startswith("misrouted_"):
        data = bench.gen_misrouted(dist[len("misrouted_"):], n, dev)
    else:
        data = bench.gen_data(dist, 0, n, dev, seeds[dist])
    job = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT), data)
elif codec == "snappy":
    data = torch.from_numpy(bench.gen_text(n * bench.CHUNK)).to(dev)
    job = bench.CodecJob(hc, hc.default_library(), "Snappy", hc.SnappyOpts(0), data)
elif codec == "cascaded":
    data = bench.gen_sorted_columns(n, dev)
    job = bench.CodecJob(hc, hc.default_library(), "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), data)
else:
    raise SystemExit("unknown row " + a.row)
job.compress(); job.decompress(); torch.cuda.synchronize()
job.verify()
tc, td = bench.time_phases(job, a.reps)
nb, cb = job.total, job.compressed_bytes()
print(f"{a.row}: chunks {job.n} ratio {nb / cb:.3f} compress {min(tc):.3f} ms {nb / min(tc) / 1e6:.1f} GB/s decompress {min(td):.3f} ms "
      f"{nb / min(td) / 1e6:.1f} GB/s algorithmic_bytes {nb + cb}", flush=True)
