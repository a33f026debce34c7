"""LZ4 parity sweep over chunk sizes and element types in the far shape (batches of 1200+ chunks of data that
compresses): every chunk against the CPU oracle, then the round trip.
   [HIPCOMP_LZ4_SHAPE=...] parity_sweep_lz4.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen
from oracle import oracle as O

hc = importlib.import_module("hipcomp-core_amd")
bad = 0
for es, dtype in ((1, 0), (2, 3), (4, 4)):
    for size in (400, 512, 1000, 4096, 10000, 65536, 70000, 200000):
        base = []
        for k in range(10):
            base.append(datagen.text_like(3000 + k, size - 3 * k))
            base.append(datagen.harness_like_int32(3100 + k, size // 4 + 1).tobytes()[: size - k])
            base.append(datagen.random_runs_int32(3200 + k, size // 4 + 1).tobytes()[: size - 2 * k])
            base.append(datagen.vocabulary_text(3300 + k, size, 64, 8))
            base.append(datagen.periodic_bytes(3400 + k, size, 3 + k, 40))
            base.append(datagen.runs_of_elements(3500 + k, size, es, 4 + 3 * k))
        base = [c[: len(c) // es * es] for c in base]
        chunks = base * 20  # 1200 chunks: more than the LDS shape holds in flight
        cap = max(len(c) for c in chunks)
        want = [O.lz4_compress(c, es, cap) for c in base]
        src = hc.batch.from_host_chunks(chunks, "cuda:0")
        codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=hc.knobs_library() if os.environ.get("HIPCOMP_LZ4_SHAPE") else None)
        comp = codec.compress(src, cap)
        torch.cuda.synchronize()
        got = comp.to_host_chunks()
        wrong = sum(1 for i in range(len(chunks)) if got[i] != want[i % len(base)])
        dec, actual, statuses = codec.decompress(comp, cap)
        ok = statuses.cpu().tolist() == [0] * len(chunks) and dec.to_host_chunks() == chunks
        print(f"es={es} size={size}: wrong={wrong} roundtrip_ok={ok}", flush=True)
        bad += wrong + (0 if ok else 1)
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
