"""Cascaded parity sweep: random partitions built to meet every bit width of the packer (values 1 .. 32 bits,
run lengths 1 .. 10 bits, odd and even, arrays that end in every position of a block of 16), all eight types and a
spread of option sets -- every partition against the CPU oracle byte for byte, then the round trip, then the
kernel decoding the ORACLE's stream.   parity_sweep_cascaded.py [rounds=6]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle as O

hc = importlib.import_module("hipcomp-core_amd")
NP = {0: np.int8, 1: np.uint8, 2: np.int16, 3: np.uint16, 4: np.int32, 5: np.uint32, 6: np.int64, 7: np.uint64}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
total = 0
for rnd in range(rounds):
    rng = np.random.default_rng(7000 + rnd)
    for t in (5, 4, 5, 3, 1, 7, 0, 2, 6):
        dt = NP[t]
        bits = 8 * np.dtype(dt).itemsize
        chunks = []
        for k in range(40):
            n = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 1000, 1023, 1024, 1025, 1500, 2047, 2048, 4096, 5000, 16384]))
            n = max(1, n + int(rng.integers(-3, 4)))
            vb = int(rng.integers(1, bits + 1))                       # bit width of the values
            mean_run = float(rng.choice([1.0, 1.3, 2.0, 5.0, 40.0, 700.0]))
            m = max(1, int(n / mean_run))
            vals = rng.integers(0, 2 ** min(vb, 63), m, dtype=np.uint64)
            if rng.random() < 0.5:
                vals = np.cumsum(vals % (2 ** min(vb, 20)), dtype=np.uint64)  # sorted: the delta layer's case
            lens = rng.geometric(1.0 / mean_run, m).astype(np.int64)
            if rng.random() < 0.3:
                lens[:] = int(rng.integers(1, 9))                   # constant run length: 0-bit length arrays
            x = np.repeat(vals, lens)[:n]
            if len(x) == 0:
                x = vals[:1]
            chunks.append(x.astype(np.uint64).astype(dt).tobytes())
        for (R, D, bp) in ((2, 1, 1), (1, 1, 1), (1, 0, 1), (0, 1, 1), (2, 2, 1), (2, 1, 0), (0, 0, 1), (3, 1, 1)):
            copts = hc.CascadedOpts(4096, t, R, D, bp)
            codec = hc.batch.Codec("Cascaded", copts)
            src = hc.batch.from_host_chunks(chunks, "cuda:0")
            mine = codec.compress(src)
            torch.cuda.synchronize()
            got = mine.to_host_chunks()
            wants = [O.cascaded_compress(c, t, R, D, bp)[0] for c in chunks]
            wrong = [i for i in range(len(chunks)) if got[i] != wants[i]]
            dec, actual, statuses = codec.decompress(mine, 65536 * 2 + 64)
            st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
            rt = [i for i in range(len(chunks)) if st[i] != 0 or ac[i] != len(chunks[i]) or dec.chunk_bytes(i, ac[i]) != chunks[i]]
            # the kernel decodes the oracle's streams
            theirs = hc.batch.from_host_chunks(wants, "cuda:0")
            dec2, actual2, statuses2 = codec.decompress(theirs, 65536 * 2 + 64)
            st2, ac2 = statuses2.cpu().tolist(), actual2.cpu().tolist()
            rt2 = [i for i in range(len(chunks)) if st2[i] != 0 or ac2[i] != len(chunks[i]) or dec2.chunk_bytes(i, ac2[i]) != chunks[i]]
            total += len(chunks)
            if wrong or rt or rt2:
                bad += len(wrong) + len(rt) + len(rt2)
                print(f"round {rnd} type {t} opts {(R, D, bp)}: wrong bytes {wrong[:5]} round trip {rt[:5]} oracle streams {rt2[:5]}", flush=True)
    print(f"round {rnd}: {total} partitions x option sets so far, bad {bad}", flush=True)
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
