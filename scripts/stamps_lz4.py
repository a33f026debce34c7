"""Diagnostic: per-segment cycle shares of the LZ4 compress window loop (HC_STAMPS build)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
hc = importlib.import_module("hipcomp-core_amd")
lib = hc.HipcompLibrary(os.path.join(ROOT, "hipcomp-core_amd/lib/libhipcomp_stamps.so"), codecs=("LZ4",))
dist, dtype = sys.argv[1], sys.argv[2]
data = bench.gen_data(dist, 5000, torch.device("cuda:0"), 0x5EED0002)
job = bench.Lz4Job(hc, lib, data, hc.hipcompType.CHAR if dtype == "char" else hc.hipcompType.INT)
job.compress(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
f = lib._dll.hipcompBatchedLZ4DebugStamps
f.argtypes = [ctypes.c_void_p]
assert f(out) == 0
v = list(out)[:8]; tot = sum(v)
names = ["0 wait words+verify, hash", "1 LDS round trip 1", "2 candidate, loads, markers (+LDS wait)", "3 decision",
         "4 insert", "5 between steps", "6 everything outside the walk", "7"]
print(dist, dtype, "total Gcycles", tot / 1e9)
for n, x in zip(names, v):
    print(f"  {n:44s} {100 * x / tot:5.1f} %   {x / (5000 * (1032 if dtype == 'char' else 258)):7.1f} ticks/window")
