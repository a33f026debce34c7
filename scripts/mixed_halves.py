import importlib, os, sys
sys.path.insert(0, '/root/repo')
import torch
import bench
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
n = 32768
text = torch.from_numpy(bench.gen_text(n * bench.CHUNK)).to(dev).view(n, bench.CHUNK)
rnd = bench.gen_data("uniform", 0, n, dev, 77).view(torch.uint8).view(n, bench.CHUNK)
def run(name, data, codec, opts):
    job = bench.CodecJob(hc, hc.default_library(), codec, opts, data.reshape(-1).contiguous())
    job.compress(); job.decompress(); torch.cuda.synchronize(); job.verify()
    tc, td = bench.time_phases(job, 4)
    print(f"{codec} {name}: compress {min(tc):.3f} ms decompress {min(td):.3f} ms", flush=True)
    return min(tc), min(td)
for codec, opts in (("Snappy", hc.SnappyOpts(0)), ("LZ4", hc.LZ4Opts(0))):
    a = run("text only", text, codec, opts)
    b = run("random only", rnd, codec, opts)
    m = run("alternating text/random (2x the chunks)", torch.stack([text, rnd], dim=1), codec, opts)
    g = run("blocks of 8 text / 8 random", torch.stack([text.view(-1, 8, bench.CHUNK), rnd.view(-1, 8, bench.CHUNK)], dim=1), codec, opts)
    print(f"   sum of the halves: compress {a[0] + b[0]:.3f} ms decompress {a[1] + b[1]:.3f} ms")
