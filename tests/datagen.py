"""Seeded input generators shared by tests, golden-vector scripts and bench.

`harness_*` follows the reference's C-API test harness
(tests/test_batch_c_api.h:232-263, 772-777): glibc srand(0)/rand(), chunk
sizes rand()%(max-min)+min ints, values (rand()%4)+300.
"""
from __future__ import annotations

import ctypes
import numpy as np

_MASK = (1 << 64) - 1


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n uint64 values of the splitmix64 stream (vectorised)."""
    idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & _MASK)
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform_int32(seed: int, n_ints: int) -> np.ndarray:
    return (splitmix64(seed, n_ints) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def harness_like_int32(seed: int, n_ints: int) -> np.ndarray:
    return (np.uint32(300) + (splitmix64(seed, n_ints) & np.uint64(3)).astype(np.uint32))


def random_runs_int32(seed: int, n_ints: int) -> np.ndarray:
    """value = run index, run length ~ U[1,16]."""
    lens = (splitmix64(seed, n_ints) % np.uint64(16)).astype(np.int64) + 1
    vals = np.repeat(np.arange(n_ints, dtype=np.uint32), lens)[:n_ints]
    return vals


def text_like(seed: int, n: int) -> bytes:
    words = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"lorem", b"ipsum",
             b"compression", b"wavefront", b"0123456789", b"|", b",", b"\n", b"AAAA", b"abcabcabc"]
    r = splitmix64(seed, n // 3 + 8) % np.uint64(len(words))
    out = bytearray()
    i = 0
    while len(out) < n:
        out += words[int(r[i])] + b" "
        i += 1
    return bytes(out[:n])


def glibc_rand(seed: int):
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    libc.rand.restype = ctypes.c_int
    return libc.rand


def harness_batches():
    """The reference harness's six batches, as lists of int32 chunk byte strings."""
    spec = [(1, 100, 101), (1, 16384, 16385), (11, 1000, 10000), (127, 10000, 16384),
            (1025, 100, 16384), (10025, 100, 1000)]
    for batch, mn, mx in spec:
        rnd = glibc_rand(0)
        sizes = [(rnd() % (mx - mn)) + mn for _ in range(batch)]
        chunks = []
        for n in sizes:
            vals = np.fromiter(((rnd() % 4) + 300 for _ in range(n)), dtype=np.int32, count=n)
            chunks.append(vals.tobytes())
        yield chunks


def edge_chunks():
    """Small adversarial set: (name, bytes)."""
    rng = np.random.default_rng(1234)
    out = [("empty", b""), ("one", b"x"), ("abcd9", b"abcd" * 9)]
    for n in (12, 13, 14, 33, 64, 65, 100, 128, 255, 256, 1021, 4096, 65535, 65536):
        out.append((f"rand4sym_{n}", bytes(rng.integers(0, 4, n, dtype=np.uint8))))
    out.append(("zeros_65536", bytes(65536)))
    out.append(("rand_65536", bytes(rng.integers(0, 256, 65536, dtype=np.uint8))))
    out.append(("ramp_i32", (np.arange(16384) >> 2).astype(np.int32).tobytes()))
    out.append(("harness_i32", harness_like_int32(7, 16384).tobytes()))
    out.append(("runs_i32", random_runs_int32(9, 16384).tobytes()))
    out.append(("text_65536", text_like(11, 65536)))
    out.append(("text_20000", text_like(12, 20000)))
    out.append(("u16_pat", (np.arange(30000) % 517).astype(np.uint16).tobytes()))
    # long matches / overlapping copies / periodic data
    out.append(("period3", (b"xyz" * 30000)[:65536]))
    out.append(("period1_then_rand", b"\x07" * 5000 + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + b"\x07" * 9000))
    return out


CASCADED_NP = {0: np.int8, 1: np.uint8, 2: np.int16, 3: np.uint16, 4: np.int32, 5: np.uint32, 6: np.int64, 7: np.uint64}


def sorted_column(seed: int, n: int) -> np.ndarray:
    """Sorted uint32 column, ~25 % repeats, increments 1..8."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, n)
    inc = np.where(g == 0, 0, rng.integers(1, 9, n))
    inc[0] = rng.integers(0, 1 << 20)
    return np.cumsum(inc).astype(np.uint32)


def predefined(values, runs, t):
    return np.repeat(np.array(values, dtype=CASCADED_NP[t]), runs).tobytes()


def cascaded_golden_inputs(t: int):
    """(name, bytes) inputs of tests/golden/cascaded_reference.json for type tag t."""
    rng = np.random.default_rng(100 + t)
    dt = CASCADED_NP[t]
    return [
        ("sorted", sorted_column(5 + t, 3000).astype(dt).tobytes()),
        ("runs", np.repeat(rng.integers(0, 100, 120), rng.integers(1, 40, 120)).astype(dt).tobytes()),
        ("zeros", np.zeros(1500, dtype=dt).tobytes()),
        ("noise", rng.integers(-100, 100, 700).astype(dt).tobytes()),
        ("incompressible", rng.integers(0, 2**31, 300).astype(dt).tobytes()),
        ("predef0", predefined([3, 9, 4, 0, 1], [1, 20, 13, 25, 6], t)),
        ("predef1", predefined([1, 2, 3, 4, 5, 6], [10, 6, 15, 1, 13, 9], t)),
    ]
