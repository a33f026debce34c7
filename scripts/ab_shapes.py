"""A/B of LZ4 launch shapes / geometries in ONE process on the same buffers (the knobs are read at
every call): ab_shapes.py --chunks N --dist harness,text --dtype char CONFIG [CONFIG ...]
CONFIG = SHAPE[:BOTH][@VARIANT] e.g. auto  far  far:4,0,2048  auto:1,7,512  auto:1,3,512@span52 (lib/libhipcomp_span52.so);
bytes are compared with the first config's."""
import argparse, importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="harness")
ap.add_argument("--dtype", default="char")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("configs", nargs="+")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
SEEDS = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}


LIBS = {}


def lib_of(cfg):
    variant = cfg.split("+")[0].partition("@")[2]
    if variant not in LIBS:
        LIBS[variant] = hc.HipcompLibrary(os.path.join(ROOT, "hipcomp-core_amd", "lib", f"libhipcomp_{variant}.so")) if variant else hc.knobs_library()  # (the knobs are read by the knobs build only)
    return LIBS[variant]


EXTRA_KEYS = set()


def set_config(cfg):
    # SHAPE[:GEOMETRY][@VARIANT][+KEY=VALUE ...]  (the +KEY=VALUE parts: environment knobs for this config only)
    cfg, *extras = cfg.split("+")
    for k in EXTRA_KEYS:
        os.environ.pop(k, None)
    for e in extras:
        k, _, v = e.partition("=")
        EXTRA_KEYS.add(k)
        os.environ[k] = v
    cfg = cfg.partition("@")[0]
    shape, _, both = cfg.partition(":")
    os.environ["HIPCOMP_LZ4_SHAPE"] = shape
    if both:
        os.environ["HIPCOMP_LZ4_GEOMETRY"] = both
    else:
        os.environ.pop("HIPCOMP_LZ4_GEOMETRY", None)


for dist in a.dist.split(","):
    if dist == "text":
        data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
    elif dist == "mixed":
        h = a.chunks // 2
        x = bench.gen_data("uniform", 0, h, dev, SEEDS["uniform"]).view(h, bench.CHUNK)
        y = bench.gen_data("harness", 0, h, dev, SEEDS["harness"]).view(h, bench.CHUNK)
        data = torch.stack([x, y], dim=1).reshape(-1).contiguous()
    else:
        data = bench.gen_data(dist, 0, a.chunks, dev, SEEDS[dist])
    for dt in a.dtype.split(","):
        t = hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT
        job = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(t), data)
        codecs = {c: hc.batch.Codec("LZ4", hc.LZ4Opts(t), lib=lib_of(c)) for c in a.configs}
        first = None
        times = {c: [] for c in a.configs}
        same = {}
        for r in range(a.rounds + 1):
            for cfg in a.configs:
                set_config(cfg)
                job.codec = codecs[cfg]
                if r == 0:
                    job.comp.data.zero_()
                    job.comp.sizes.zero_()
                    job.compress(); torch.cuda.synchronize()
                    snap = (job.comp.sizes.clone(), job.comp.data.clone())
                    if first is None:
                        first = snap
                        job.decompress(); torch.cuda.synchronize(); job.verify()
                    same[cfg] = bool(torch.equal(snap[0], first[0])) and bool(torch.equal(snap[1], first[1]))
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); job.compress(); e1.record(); torch.cuda.synchronize()
                    times[cfg].append(e0.elapsed_time(e1))
        nb = job.total
        for cfg in a.configs:
            print(f"{dist:8s} {dt:4s} n={job.n:6d} {cfg:22s} compress min {min(times[cfg]):8.3f} ms {nb / min(times[cfg]) / 1e6:8.1f} GB/s"
                  f"  med {statistics.median(times[cfg]):8.3f}  same_bytes={same[cfg]}", flush=True)
        del job
    del data
    torch.cuda.empty_cache()
