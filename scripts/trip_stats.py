"""What the trips of the lean far form do on a data kind (measurement build:
make -C hipcomp-core_amd/csrc VARIANT=stats EXTRA=-DHC_TRIP_STATS): trip_stats.py --chunks N --dist text,harness [--config far:4,0,2048]"""
import argparse, ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=2000)
ap.add_argument("--dist", default="text")
ap.add_argument("--dtype", default="char")
ap.add_argument("--config", default="far")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
path = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp_stats.so")
lib = hc.HipcompLibrary(path)
dll = ctypes.CDLL(path)
dev = torch.device("cuda:0")
shape, _, both = a.config.partition(":")
os.environ["HIPCOMP_LZ4_SHAPE"] = shape
if both:
    os.environ["HIPCOMP_LZ4_GEOMETRY"] = both
names = ["trips", "sequences taken in trips", "trips ended by a match >= 16 bytes", "trips that took nothing", "pick: no match lane left in the span",
         "pick: clash (slot shared / stale)", "windows of the general path", "sequences followed without a look", "lanes moved (sum)",
         "pick: more literals than a token holds", "wide form: sequences taken", "wide form: ... with the length off the window",
         "runs trip: tried", "runs trip: no lane with a candidate", "runs trip: took sequences", "runs trip: sequences taken"]
for dist in a.dist.split(","):
    if dist == "text":
        data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
    else:
        data = bench.gen_data(dist, 0, a.chunks, dev, {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}[dist])
    t = hc.hipcompType.CHAR if a.dtype == "char" else hc.hipcompType.INT
    job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(t), data)
    buf = (ctypes.c_uint32 * 16)()
    assert dll.hipcompBatchedLZ4DebugTripStats(buf, 1) == 0
    job.compress(); torch.cuda.synchronize()
    assert dll.hipcompBatchedLZ4DebugTripStats(buf, 1) == 0
    v = list(buf)
    print(f"== {dist} {a.dtype} {a.config}: {job.n} chunks, ratio {job.total / job.compressed_bytes():.3f}")
    for i, nme in enumerate(names):
        print(f"   {nme:48s} {v[i]:12d}   per chunk {v[i] / job.n:10.1f}")
    if v[0]:
        print(f"   sequences per trip {v[1] / v[0]:.2f}, lanes per trip {v[8] / v[0]:.1f}")
    del job, data
