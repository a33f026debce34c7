"""Where a chunk's time goes in the LZ4 mix kernel (diagnostic build: make -C hipcomp-core_amd/csrc VARIANT=mixstamps
EXTRA=-DHC_MIX_STAMPS): mix_stamps.py [--chunks N] [--dtype char|int]"""
import argparse, ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=100000)
ap.add_argument("--dtype", default="char,int")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
path = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp_mixstamps.so")
lib = hc.HipcompLibrary(path)
dll = ctypes.CDLL(path)
dev = torch.device("cuda:0")
names = ["LDS table init", "walk_run (blocks of 4 windows)", "literal run at the end of the chunk (incl. waiting for its stores)", "emit_match (literals + match of a sequence)", "whole chunk"]
data = bench.gen_data("uniform", 0, a.chunks, dev, 0x5EED0002)
for dt in a.dtype.split(","):
    job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT), data)
    job.compress(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    assert dll.hipcompBatchedLZ4DebugMixStamps(buf, 1) == 0
    job.compress(); torch.cuda.synchronize()
    assert dll.hipcompBatchedLZ4DebugMixStamps(buf, 1) == 0
    n = max(buf[5], 1)
    print(f"uniform as {dt}: {n} chunks; s_memtime ticks (shader clock cycles) per chunk per wave:")
    for k, nme in enumerate(names):
        print(f"   {nme:70s} {buf[k] / n:10.0f} cycles  {buf[k] / n / 2400:8.2f} us at 2.4 GHz")
    del job
