import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
kind = sys.argv[1]
d = bench.gen_misrouted(kind, 16384, dev)
job = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d)
for _ in range(4):
    job.compress()
torch.cuda.synchronize()
