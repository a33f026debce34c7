"""The Cascaded encoder on the config-3 columns with one build of the library, compress only (the stage
builds of scripts/pmc_cascaded_stages.sh do not write streams):  casc_stage_run.py LIB [partitions] [reps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
lib = hc.HipcompLibrary(os.path.join(ROOT, sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
data = bench.gen_sorted_columns(n, dev)
job = bench.CodecJob(hc, lib, "Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1), data)
job.compress(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job.compress(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{sys.argv[1]}: {n} partitions, compress {min(ts) * 1e3:.3f} ms  {job.total / min(ts) / 1e9:.1f} GB/s", flush=True)
