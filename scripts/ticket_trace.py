"""The mix kernel's ticket counter over time (knobs build, HIPCOMP_PREFETCH_TRACE=1): the prefetch kernel's
first wave notes {100 MHz clock, counter} whenever the counter moved, into the dense class's list in the temp
buffer.  Prints the counter's steps: are the 1024 resident waves in lock-step (bursts of tickets) or spread?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HIPCOMP_PREFETCH_TRACE"] = "1"
import torch, numpy as np
import bench
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dt = sys.argv[2] if len(sys.argv) > 2 else "char"
data = bench.gen_data("uniform", 0, n, dev, 0x5EED0002)
job = bench.CodecJob(hc, hc.knobs_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT), data)
job.compress(); torch.cuda.synchronize()
job.temp[256 + 4 * n: 256 + 8 * n].zero_()
job.compress(); torch.cuda.synchronize()
base = job.temp.data_ptr() % 4
assert base == 0
words = job.temp[256 + 4 * n: 256 + 8 * n].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
pairs = words.reshape(-1, 2)
pairs = pairs[pairs[:, 1] > 0]
t0 = pairs[0, 0]
us = ((pairs[:, 0] - t0) & 0xFFFFFFFF) / 100.0
tk = pairs[:, 1]
print(f"{len(pairs)} samples over {us[-1]:.0f} us, counter {tk[0]} .. {tk[-1]}")
# tickets per 10 us bin over the first 1.2 ms and around the middle
for lo in (0.0, us[-1] / 2):
    print(f"-- tickets taken per 10 us from {lo:.0f} us on")
    row = []
    for b in range(60):
        a, z = lo + 10 * b, lo + 10 * (b + 1)
        sel = (us >= a) & (us < z)
        row.append(int(tk[sel].max() - tk[sel].min()) if sel.any() else 0)
    print(" ".join(f"{v:4d}" for v in row))
