"""quick_hlif.py [--lib path]: the HLIF LZ4 manager row of bench.py alone (6.55 GB uniform), best of 3."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
if a.lib:
    lib = hc.HipcompLibrary(os.path.join(ROOT, a.lib))
    hc.default_library = lambda: lib
    bench_default = lib
d = bench.gen_data("uniform", 0, 100000, torch.device("cuda:0"), 0x5EED0002)
r = bench.measure_hlif(hc, d, reps=3)
print(f"hlif compress {r['compress_ms']:.2f} ms {r['compress_GBps']:.1f} GB/s | decompress {r['decompress_ms']:.2f} ms {r['decompress_GBps']:.1f} GB/s")
