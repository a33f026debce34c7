"""Repeats the compress of a mixed batch (text, harness data, runs, sparse repeats, random bytes; ragged lengths)
and compares every chunk with the CPU oracle each time: a race shows as an occasional wrong chunk.
   [HIPCOMP_LZ4_SHAPE=mix|far] stress_lz4_parity.py [--reps N] [--dtype char|int]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import datagen
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--dtype", default="int")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
es, dtype = (4, 4) if a.dtype == "int" else (1, 0)
rng = np.random.default_rng(5)
base = []
for k in range(12):
    base.append(datagen.text_like(200 + k, 65536 - 97 * k))
    base.append(datagen.harness_like_int32(300 + k, 16384 - 3 * k).tobytes())
    base.append(datagen.random_runs_int32(400 + k, 16384 - 5 * k).tobytes())
for k, (every, length) in enumerate([(200, 4), (200, 9), (700, 5), (3000, 40), (61, 4), (64, 6), (5000, 300), (129, 4)]):
    for n in (65536, 65535 - 7 * k, 20000 + 13 * k):
        base.append(datagen.sparse_repeats(100 + k, n, every, length))
base += [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (65536, 65536, 40000, 257)]
base = [c[: len(c) // es * es] for c in base]
chunks = base * 24
want = [O.lz4_compress(c, es, 65536) for c in base]
src = hc.batch.from_host_chunks(chunks, "cuda:0")
codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=hc.knobs_library() if os.environ.get("HIPCOMP_LZ4_SHAPE") else None)
bad = 0
for r in range(a.reps):
    got = codec.compress(src, 65536).to_host_chunks()
    wrong = [i for i in range(len(chunks)) if got[i] != want[i % len(base)]]
    if wrong:
        bad += 1
        print(f"rep {r}: {len(wrong)} wrong chunks, first {wrong[:5]} (base {[w % len(base) for w in wrong[:5]]})", flush=True)
print(f"{a.reps} reps x {len(chunks)} chunks ({a.dtype}, shape {os.environ.get('HIPCOMP_LZ4_SHAPE', 'auto')}): {bad} reps with wrong chunks")
sys.exit(1 if bad else 0)
