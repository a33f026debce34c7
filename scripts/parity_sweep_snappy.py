"""Snappy parity sweep over chunk sizes (data that compresses): every chunk against the CPU oracle, then the
round trip.   parity_sweep_snappy.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen
from oracle import oracle as O

hc = importlib.import_module("hipcomp-core_amd")
bad = 0
for size in (100, 143, 144, 145, 300, 1000, 4096, 10000, 65536, 70000, 200000):
    chunks = []
    for k in range(12):
        chunks.append(datagen.text_like(4000 + k, size - k))
        chunks.append(datagen.harness_like_int32(4100 + k, size // 4 + 1).tobytes()[: size - k])
        chunks.append(datagen.random_runs_int32(4200 + k, size // 4 + 1).tobytes()[: size - 2 * k])
        chunks.append(datagen.vocabulary_text(4300 + k, size, 64, 8))
        chunks.append(datagen.periodic_bytes(4400 + k, size, 2 + k, 40))
        chunks.append(datagen.small_alphabet_bytes(4500 + k, size, 2 + k))
    cap = max(len(c) for c in chunks)
    want = [O.snappy_compress(c) for c in chunks]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("Snappy")
    comp = codec.compress(src)
    torch.cuda.synchronize()
    got = comp.to_host_chunks()
    wrong = sum(1 for i in range(len(chunks)) if got[i] != want[i])
    dec, actual, statuses = codec.decompress(comp, cap)
    ok = statuses.cpu().tolist() == [0] * len(chunks) and dec.to_host_chunks() == chunks
    print(f"size={size}: wrong={wrong} roundtrip_ok={ok}", flush=True)
    bad += wrong + (0 if ok else 1)
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
