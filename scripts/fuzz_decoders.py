"""Damaged LZ4 and Snappy streams (bytes changed, inserted, removed, streams cut short) decoded on the GPU and by the
CPU oracle: status, reported size and -- on success -- bytes must agree.   fuzz_decoders.py [--per-source N]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import datagen
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--per-source", type=int, default=400)
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")


def damaged(rng, good, n):
    out = [good]
    for k in range(n):
        b = bytearray(good)
        kind = k % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            del b[int(rng.integers(0, len(b)))]
        elif kind == 2:
            b.insert(int(rng.integers(0, len(b) + 1)), int(rng.integers(0, 256)))
        elif kind == 3:
            b = b[: int(rng.integers(1, len(b)))]
        else:  # a changed offset byte pair somewhere: near / far / zero offsets
            at = int(rng.integers(0, max(1, len(b) - 2)))
            v = int(rng.choice([0, 1, 2, 3, 4, 7, 63, 64, 65, 255, 256, 4000, 65535]))
            b[at] = v & 0xFF
            b[at + 1] = v >> 8
        out.append(bytes(b))
    return out


sources = [datagen.text_like(31, 6000), datagen.harness_like_int32(32, 1500).tobytes(),
           datagen.random_runs_int32(33, 1500).tobytes(), datagen.vocabulary_text(34, 6000, 64, 8),
           datagen.periodic_bytes(35, 5000, 3, 40), datagen.small_alphabet_bytes(36, 5000, 3),
           datagen.tpch_lineitem_text(37, 6000)]
bad = 0
rng = np.random.default_rng(2025)
for codec_name, comp_fn, dec_fn in (("LZ4", lambda s: O.lz4_compress(s, 1, 65536), O.lz4_decompress),
                                    ("Snappy", O.snappy_compress, O.snappy_decompress)):
    streams = []
    for src in sources:
        streams += damaged(rng, comp_fn(src), a.per_source)
    for cap in (6000, 3500):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        dec, actual, statuses = hc.batch.Codec(codec_name).decompress(comp, cap)
        torch.cuda.synchronize()
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        wrong = 0
        for i, s in enumerate(streams):
            ost, obytes = dec_fn(s, cap)
            if st[i] != ost or ac[i] != len(obytes) or (ost == 0 and dec.chunk_bytes(i, ac[i]) != obytes):
                wrong += 1
                if wrong <= 3:
                    print("MISMATCH", codec_name, cap, i, st[i], ost, ac[i], len(obytes))
        print(f"{codec_name} cap={cap}: {len(streams)} streams, {wrong} differ", flush=True)
        bad += wrong
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
