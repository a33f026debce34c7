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


# ---- TPC-H lineitem-like text (BASELINE.json configs[3]) --------------------
# No dbgen and no network here: a deterministic generator that follows the
# lineitem column grammar of the TPC-H specification (clause 4.2.3): keys,
# quantities, decimal prices, flags, three dates, ship instructions / modes and
# a comment built from the spec's word lists, fields separated by "|".
_TPCH_INSTRUCT = ["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"]
_TPCH_MODES = ["REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"]
_TPCH_WORDS = ("furiously sly carefully blithely quickly fluffily slyly quietly ruthlessly thinly closely doggedly "
               "daringly bravely stealthily permanently enticingly idly busily regular final ironic even bold silent "
               "special pending unusual express packages requests accounts deposits foxes ideas theodolites pinto beans "
               "instructions dependencies excuses platelets asymptotes courts dolphins multipliers sauternes warthogs "
               "frets dinos attainments somas Tiresias patterns forges braids hockey players frays warhorses dugouts "
               "notornis epitaphs pearls tithes waters orbits gifts sheaves depths sentiments decoys realms pains "
               "grouches escapades sleep wake are cajole haggle nag use boost affix detect integrate maintain nod was "
               "lose sublate solve thrash promise engage hinder print x-ray breach eat grow impress mold poach serve run "
               "dazzle snooze doze unwind kindle play hang believe doubt about above according to across after against "
               "along alongside of among around at atop before behind beneath beside besides between beyond by despite "
               "during except for from in place of inside instead of into near of on outside over past since through "
               "throughout to toward under until up upon without with within").split()


def tpch_lineitem_text(seed: int, n_bytes: int) -> bytes:
    rng = np.random.default_rng(seed)
    rows = n_bytes // 100 + 64
    okey = np.cumsum(rng.integers(0, 2, rows) * rng.integers(1, 5, rows)) + 1
    line = np.ones(rows, dtype=np.int64)
    for i in range(1, 8):  # line numbers 1..7 within an order
        same = np.zeros(rows, dtype=bool)
        same[i:] = okey[i:] == okey[:-i]
        line = np.where(same, np.maximum(line, i + 1), line)
    part = rng.integers(1, 200001, rows)
    supp = rng.integers(1, 10001, rows)
    qty = rng.integers(1, 51, rows)
    price = qty * (90000 + (part // 10) % 20001 + 100 * (part % 1000))  # cents
    disc = rng.integers(0, 11, rows)
    tax = rng.integers(0, 9, rows)
    day0 = rng.integers(0, 2406, rows)  # days since 1992-01-02
    base = np.datetime64("1992-01-02")
    ship = (base + day0).astype(str)
    commit = (base + day0 + rng.integers(-60, 61, rows)).astype(str)
    receipt = (base + day0 + rng.integers(1, 31, rows)).astype(str)
    rflag = np.where(day0 > 1260, "N", np.where(rng.integers(0, 2, rows) == 0, "R", "A"))
    lstat = np.where(day0 > 1260, "O", "F")
    instr = np.array(_TPCH_INSTRUCT)[rng.integers(0, 4, rows)]
    mode = np.array(_TPCH_MODES)[rng.integers(0, 7, rows)]
    words = np.array(_TPCH_WORDS)
    nw = rng.integers(2, 7, rows)
    comment = words[rng.integers(0, len(words), rows)]
    for k in range(1, 6):
        extra = np.char.add(" ", words[rng.integers(0, len(words), rows)])
        comment = np.where(nw > k, np.char.add(comment, extra), comment)
    def dec(v):  # cents -> "123.45"
        return np.char.add(np.char.add((v // 100).astype(str), "."), np.char.zfill((v % 100).astype(str), 2))
    cols = [okey.astype(str), part.astype(str), supp.astype(str), line.astype(str), qty.astype(str), dec(price),
            np.char.add("0.", np.char.zfill(disc.astype(str), 2)), np.char.add("0.", np.char.zfill(tax.astype(str), 2)),
            rflag, lstat, ship, commit, receipt, instr, mode, comment]
    row = cols[0]
    for c in cols[1:]:
        row = np.char.add(np.char.add(row, "|"), c)
    text = "\n".join(row.tolist()).encode() + b"\n"
    while len(text) < n_bytes:
        text += text
    return text[:n_bytes]


def sparse_repeats(seed: int, n: int, every: int, length: int) -> bytes:
    """Incompressible bytes with a copy of `length` earlier bytes planted about every `every`
    bytes: long match-less stretches (the LZ4 encoder's pipelined walk) that end in a match at
    an arbitrary lane of an arbitrary window (its roll-back)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    pos = 0
    while True:
        pos += int(rng.integers(every // 2, every * 3 // 2 + 1))
        if pos + length >= n:
            break
        back = int(rng.integers(1, min(pos, 65535) + 1))
        src = pos - back
        for k in range(length):  # byte by byte: overlapping copies stay LZ-like
            a[pos + k] = a[src + k]
        pos += length
    return a.tobytes()


def periodic_bytes(seed: int, n: int, period: int, mutate_every: int) -> bytes:
    """A random pattern of `period` bytes repeated, one byte changed about every `mutate_every`
    bytes: matches at the distance of the period (shorter than a window for small periods),
    literal runs of one byte, lanes of a window that share hash slots."""
    r = splitmix64(seed, period + 2 * (n // max(mutate_every, 1) + 2))
    pat = (r[:period] & np.uint64(0xFF)).astype(np.uint8)
    out = np.tile(pat, n // period + 1)[:n].copy()
    at = 0
    for k in range(n // max(mutate_every, 1)):
        at += 1 + int(r[period + 2 * k] % np.uint64(2 * mutate_every))
        if at >= n:
            break
        out[at] = np.uint8(int(r[period + 2 * k + 1]) & 0xFF)
    return out.tobytes()


def small_alphabet_bytes(seed: int, n: int, symbols: int) -> bytes:
    """Random bytes out of `symbols` values: short matches everywhere, equal words inside a window."""
    return ((splitmix64(seed, n) % np.uint64(symbols)).astype(np.uint8) + np.uint8(0x41)).tobytes()


def vocabulary_text(seed: int, n: int, words: int, longest: int) -> bytes:
    """Random words out of a vocabulary of `words` (3..longest letters), one separator byte:
    sequences of a few literals and a short match, candidates near and far."""
    r = splitmix64(seed, words * (longest + 1) + n // 3 + 8)
    vocab = []
    k = 0
    for _ in range(words):
        ln = 3 + int(r[k] % np.uint64(longest - 2)); k += 1
        vocab.append(bytes(0x61 + int(r[k + j] % np.uint64(26)) for j in range(ln)))
        k += ln
    parts, total = [], 0
    while total < n:
        w = vocab[int(r[k % len(r)] % np.uint64(words))]; k += 1
        parts.append(w + b" ")
        total += len(w) + 1
    return b"".join(parts)[:n]


def trip_corner_chunks():
    """Chunks for the encoders' several-elements-per-trip paths (LZ4 far, Snappy): 64 of them."""
    out = []
    for i, p in enumerate((1, 2, 3, 4, 5, 7, 8, 11, 12, 16, 19, 24, 31, 40)):
        out.append(periodic_bytes(900 + i, 65536 - 13 * i, p, 37 + 11 * i))
        out.append(periodic_bytes(950 + i, 65536 - 5 * i, p, 9 + i))
    for i, s in enumerate((2, 3, 4, 5, 8, 16)):
        out.append(small_alphabet_bytes(1000 + i, 65536 - 31 * i, s))
        out.append(small_alphabet_bytes(1050 + i, 30000 + 977 * i, s))
    for i, (w, l) in enumerate(((4, 5), (16, 6), (64, 8), (256, 9), (1024, 12), (4096, 7))):
        out.append(vocabulary_text(1100 + i, 65536 - 3 * i, w, l))
        out.append(vocabulary_text(1150 + i, 65536 - 101 * i, w, l))
    for i in range(6):
        out.append(text_like(1200 + i, 65536 - i))
        out.append(tpch_lineitem_text(1250 + i, 65536 - 7 * i))
    assert len(out) == 64
    return out


def runs_of_elements(seed: int, n_bytes: int, elem_size: int, longest: int = 16) -> bytes:
    """Run-length data whose values are of the element's size: value = run index (wrapping in
    the element), run lengths ~ U[1, longest] elements."""
    n = n_bytes // elem_size + longest
    r = splitmix64(seed, n)
    lengths = (r % np.uint64(longest)).astype(np.int64) + 1
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[elem_size]
    values = np.arange(len(lengths), dtype=np.uint64).astype(dt)
    return np.repeat(values, lengths)[: n_bytes // elem_size].tobytes()


def lane63_case(elem_size: int, layout: int, seed: int = 0):
    """A chunk crafted for the LZ4 encoder's several-sequences trips at a span of 64 lanes (lz4_far.hiph,
    pick() and the chain walk): a trip whose last sequence ends BEHIND lane 63 while lane 63 itself has a
    table match.  Elements of `elem_size` bytes.
      layout 1  trip = [1 literal][k short matches][l literals][match A from lane p, q elements, p + q > 64]:
                lane 63's word lies inside A, whose first occurrence went into the table -- bit 63 of the
                trip's match mask is set while the next window starts beyond lane 63 (the general walk)
      layout 2  trip = [k short matches][match A ending at lane 62][match B AT lane 63]: a chain of
                sequences without literals that runs through lane 63 (the chain walk)
    The strings are planted in a match-less prefix first (none of their elements at window lane 31,
    which the reference never inserts), each followed by an element of its own.
    Returns (bytes, expected sequences as (literal bytes, match bytes) behind the first one)."""
    S = elem_size
    rng = np.random.default_rng(1000 * S + 10 * layout + seed)
    NV = {1: 61, 2: 63, 4: 64}[S]
    used = set()

    def uniq(n):
        out = []
        while len(out) < n:
            v = int(rng.integers(1, 1 << (8 * S)))
            if S == 1 or v not in used:
                used.add(v)
                out.append(v)
        return out
    m = {1: 8, 2: 4, 4: 2}[S]  # a short match: 8 bytes
    if layout == 1:
        k, l, q = {1: (6, 3, 15), 2: (14, 1, 7), 4: (30, 1, 3)}[S]
    else:
        k, q, r = {1: (6, 15, 8), 2: (14, 7, 4), 4: (30, 3, 2)}[S]
    strings = [uniq(m) for _ in range(k + 1)] + [uniq(q)] + ([uniq(r)] if layout == 2 else [])
    pre = []
    for s_ in strings:
        pre += uniq(3)
        while any((len(pre) + j) % NV == 31 for j in range(len(s_))):
            pre += uniq(1)
        pre += s_ + uniq(1)
    pre += uniq(2 * NV)
    while len(pre) % NV:
        pre += uniq(1)
    Ms, A = strings[:k + 1], strings[k + 1]
    region = list(Ms[0])
    if layout == 1:
        region += uniq(1)
        for M in Ms[1:]:
            region += M
        region += uniq(l) + A + uniq(40)
        want = [(S, m * S)] + [(0, m * S)] * (k - 1) + [(l * S, q * S)]
    else:
        for M in Ms[1:]:
            region += M
        region += A + strings[k + 2] + uniq(40)
        want = [(0, m * S)] * k + [(0, q * S), (0, r * S)]
    els = pre + region + uniq(200)
    return np.array(els, dtype={1: np.uint8, 2: np.uint16, 4: np.uint32}[S]).tobytes(), want
