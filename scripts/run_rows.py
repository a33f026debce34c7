"""Several bench rows' workloads in ONE process (for rocprofv3 --pmc passes: the counters come per dispatch
and the rows' kernels differ by name, so one process start serves them all):
   run_rows.py lz4/text/char/20000 snappy/text/20000 lz4/harness/char/20000 cascaded/sorted/20000 [--reps N]
Row keys as bench.py writes them into extra_keys[].row; prints one line per row (chunks, ratio, timings and
the LZ4 sequences / Snappy elements per chunk that the per-trip reductions divide by)."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("rows", nargs="+")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
seeds = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
lib = hc.HipcompLibrary(os.path.join(ROOT, a.lib)) if a.lib else hc.default_library()
text = {}


def text_of(n):
    if n not in text:
        text.clear()
        text[n] = torch.from_numpy(bench.gen_text(n * bench.CHUNK)).to(dev)
    return text[n]


for row in a.rows:
    parts = row.split("/")
    codec, n = parts[0], int(parts[-1])
    if codec == "lz4":
        dist, dt = parts[1], parts[2]
        if dist == "text":
            data = text_of(n)
        elif dist == "mixed":
            data = bench.gen_mixed(n, dev)
        elif dist.startswith("misrouted_"):
            data = bench.gen_misrouted(dist[len("misrouted_"):], n, dev)
        else:
            data = bench.gen_data(dist, 0, n, dev, seeds[dist])
        job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT), data)
    elif codec == "snappy":
        data = text_of(n)
        job = bench.CodecJob(hc, lib, "Snappy", hc.SnappyOpts(0), data)
    elif codec == "cascaded":
        data = bench.gen_sorted_columns(n, dev)
        job = bench.CodecJob(hc, lib, "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), data)
    else:
        raise SystemExit("unknown row " + row)
    job.compress(); job.decompress(); torch.cuda.synchronize()
    job.verify()
    tc, td = bench.time_phases(job, a.reps)
    nb, cb = job.total, job.compressed_bytes()
    print(f"{row}: chunks {job.n} ratio {nb / cb:.3f} compress {min(tc):.3f} ms {nb / min(tc) / 1e6:.1f} GB/s decompress {min(td):.3f} ms "
          f"{nb / min(td) / 1e6:.1f} GB/s algorithmic_bytes {nb + cb}", flush=True)
    del job, data
    torch.cuda.empty_cache()
