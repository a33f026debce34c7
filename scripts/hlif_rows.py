import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
LIB = sys.argv[2] if len(sys.argv) > 2 else None   # (another build of the library, for an A/B of the managers)
text = torch.from_numpy(bench.gen_text(n * bench.CHUNK)).to(dev)
print(json.dumps(bench.measure_hlif(hc, text, "Snappy", lib_path=LIB)))
job = bench.CodecJob(hc, hc.default_library(), "Snappy", hc.SnappyOpts(0), text)
import time
for name, f in (("compress", job.compress), ("decompress", job.decompress)):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(3):
        t0=time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print("batched snappy", name, text.numel()/min(ts)/1e9)
del job, text
cols = bench.gen_sorted_columns(n, dev).view(torch.uint8)
print(json.dumps(bench.measure_hlif(hc, cols, "Cascaded", lib_path=LIB)))
job = bench.CodecJob(hc, hc.default_library(), "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), cols)
for name, f in (("compress", job.compress), ("decompress", job.decompress)):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(3):
        t0=time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print("batched cascaded", name, cols.numel()/min(ts)/1e9)
d = bench.gen_data("uniform", 0, n, dev, 0x5EED0002).view(torch.uint8)
print(json.dumps(bench.measure_hlif(hc, d, "LZ4", lib_path=LIB)))
