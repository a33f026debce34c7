import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
for kind in bench.MISROUTED_KINDS + ("text", "uniform"):
    if kind == "text":
        d = torch.from_numpy(bench.gen_text(16384 * bench.CHUNK)).to(dev)
    elif kind == "uniform":
        d = bench.gen_data("uniform", 0, 16384, dev, 0x5EED0007).view(torch.uint8)
    else:
        d = bench.gen_misrouted(kind, 16384, dev)
    job = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d)
    job.compress(); job.decompress(); torch.cuda.synchronize(); job.verify()
    tc, td = bench.time_phases(job, 3)
    print(f"{kind:24s} compress {job.total/min(tc)/1e6:7.1f} GB/s  decompress {job.total/min(td)/1e6:7.1f} GB/s  ratio {job.total/job.compressed_bytes():.3f}", flush=True)
    del job, d
