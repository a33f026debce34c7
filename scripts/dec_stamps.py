"""Where a several-sequences step of the LZ4 decoder spends its cycles (diagnostic build:
make -C hipcomp-core_amd/csrc VARIANT=stamps EXTRA=-DHC_DEC_STAMPS): dec_stamps.py [--chunks N] [--dist harness,text,runs]"""
import argparse, ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="harness,text,runs")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
path = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp_stamps.so")
lib = hc.HipcompLibrary(path)
dll = ctypes.CDLL(path)
dev = torch.device("cuda:0")
SEEDS = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
names = ["window (ensure)", "tokens + walk", "output lanes, match source load issued, pointer jumping", "wait for the source bytes", "store issue"]
for dist in a.dist.split(","):
    data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev) if dist == "text" else bench.gen_data(dist, 0, a.chunks, dev, SEEDS[dist])
    job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), data)
    job.compress(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    job.decompress(); torch.cuda.synchronize()
    assert dll.hipcompBatchedLZ4DebugDecodeStamps(buf, 1) == 0
    job.decompress(); torch.cuda.synchronize()
    assert dll.hipcompBatchedLZ4DebugDecodeStamps(buf, 1) == 0
    steps = max(buf[5], 1)
    print(f"{dist}: {steps / a.chunks:.0f} several-sequences steps per chunk, {buf[6] / steps:.1f} output bytes per step; cycles per step per wave:")
    for k, nme in enumerate(names):
        print(f"   {nme:60s} {buf[k] / steps:8.0f}")
    print(f"   {'sum':60s} {sum(buf[k] for k in range(5)) / steps:8.0f}")
    del job, data
    torch.cuda.empty_cache()
