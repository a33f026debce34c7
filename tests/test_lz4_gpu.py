"""LZ4 parity on the GPU, through the C ABI: compressed bytes vs the CPU
oracle and vs the reference's own build (oracle/_ref; when it is absent the
test is reported as skipped, with the reason, after its oracle asserts have
passed), round trips, the reference harness's batches and its error-path checks
(reference tests/test_batch_c_api.h:225-790).  The encoder tests run once per
launch shape (conftest.lz4_shape): the sampler's choice and every forced shape,
also on data the shape would never be chosen for."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import datagen
from conftest import compare_with_reference

pytestmark = pytest.mark.gpu

TYPES = [("CHAR", 0, 1), ("USHORT", 3, 2), ("INT", 4, 4)]

_ORACLE_CACHE = {}


def _want(oracle, c: bytes, es: int, max_chunk: int, **kw) -> bytes:
    """oracle.lz4_compress, remembered (the same chunks come back once per shape)"""
    key = (hashlib.sha1(c).digest(), len(c), es, max_chunk, tuple(sorted(kw.items())))
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.lz4_compress(c, es, max_chunk, **kw)
    return _ORACLE_CACHE[key]


def _header(temp):
    """the launcher's 64-word header at the head of the temp buffer (lz4_launch.hpp): ticket counters
    and list lengths per class {mix, dense, sparse, wide}, totals of the samples"""
    import torch
    h = temp[:64].view(torch.int32).cpu().tolist()
    return h[0:4], h[4:8], h[8:11]


def _compress(hc, chunks, dtype, max_chunk, lib=None):
    import torch
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    out = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=lib).compress(src, max_chunk)
    torch.cuda.synchronize()
    return src, out


def _round_trip(hc, comp, chunks, dtype, cap=65536):
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype))
    dec, actual, statuses = codec.decompress(comp, cap)
    assert statuses.cpu().tolist() == [0] * len(chunks)
    assert dec.to_host_chunks() == chunks
    return codec


def _reference_agrees(hc, chunks, dtype, max_chunk, want, what):
    """a check for compare_with_reference: the reference build's bytes == want"""
    def check(reflib):
        _, ref = _compress(hc, chunks, dtype, max_chunk, lib=reflib)
        refgot = ref.to_host_chunks()
        for i in range(len(chunks)):
            assert refgot[i] == want[i], f"{what} chunk {i}: oracle != reference build"
    return check


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_edge_chunks_bit_exact(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    named = datagen.edge_chunks()
    # typed modes need sizeof(T)-aligned starts (true: 16-B stride) -- any length is legal
    chunks = [c for _, c in named]
    wants = {}
    for max_chunk in (65536, 0, 1000):
        src, mine = _compress(hc, chunks, dtype, max_chunk)
        got = mine.to_host_chunks()
        wants[max_chunk] = [_want(oracle, c, es, max_chunk) for c in chunks]
        for i, (name, c) in enumerate(named):
            assert got[i] == wants[max_chunk][i], f"{name} {tname} max_chunk={max_chunk} shape={lz4_shape}: kernel != oracle"
        codec = _round_trip(hc, mine, chunks, dtype)
        assert codec.get_decompress_size(mine).cpu().tolist() == [len(c) for c in chunks]

    def check(reflib):
        for max_chunk in (65536, 0, 1000):
            _reference_agrees(hc, chunks, dtype, max_chunk, wants[max_chunk], f"{tname} max_chunk={max_chunk}")(reflib)
    compare_with_reference(reflib, "edge chunks", check)


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_sparse_matches_walk_and_rollback(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """Long match-less stretches that end in a match: the encoder's pipelined
    walk, its roll-back (match in the older of the two windows in flight, at
    any lane), its exits at the last full window, and the re-arming after a
    match.  More chunks than one workgroup holds, of different sizes, so that
    waves draw tickets at different times.  (Forced far shapes: their general,
    match-less path on data they are never chosen for.)"""
    chunks = []
    for k, (every, length) in enumerate([(200, 4), (200, 9), (700, 5), (3000, 40), (61, 4), (64, 6), (5000, 300), (129, 4)]):
        for n in (65536, 65535 - 7 * k, 20000 + 13 * k, 257 + k):
            chunks.append(datagen.sparse_repeats(100 + k, n, every, length))
    chunks.append(bytes(np.random.default_rng(9).integers(0, 256, 65536, dtype=np.uint8)))  # no match at all
    chunks = [c[: len(c) // es * es] if es > 1 else c for c in chunks]
    src, mine = _compress(hc, chunks, dtype, 65536)
    got = mine.to_host_chunks()
    want = [_want(oracle, c, es, 65536) for c in chunks]
    for i in range(len(chunks)):
        assert got[i] == want[i], f"chunk {i} {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "sparse matches", _reference_agrees(hc, chunks, dtype, 65536, want, tname))


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_compressible_batches_take_the_far_shape(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """A batch of data that compresses: the routing kernel sends nearly all of
    it to the far shapes (hash tables in the temp buffer or, for the waves that
    have one, in LDS).  Checked: the lists it left in the temp buffer, and every chunk's
    bytes against the oracle (48 distinct chunks, each 32 times: text, the
    harness's data, runs, sparse repeats, ragged lengths, an empty chunk, and a
    few chunks of random bytes, which take that kernel's match-less path) and
    the reference build."""
    import torch
    rng = np.random.default_rng(5)
    base = []
    for k in range(12):
        base.append(datagen.text_like(200 + k, 65536 - 97 * k))
        base.append(datagen.harness_like_int32(300 + k, 16384 - 3 * k).tobytes())
        base.append(datagen.random_runs_int32(400 + k, 16384 - 5 * k).tobytes())
    for k in range(6):
        base.append(datagen.sparse_repeats(500 + k, 65536 - 8 * k, 90 + 40 * k, 5 + k))
    base += [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (65536, 65536, 40000, 257)]
    base += [b"", datagen.text_like(7, 12 * es)]
    base = [c[: len(c) // es * es] for c in base]
    assert len(base) == 48
    chunks = base * 32
    want = [_want(oracle, c, es, 65536) for c in base]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype))
    dst = hc.batch.alloc_batch(src.n, codec.max_output_chunk_size(65536), src.device)
    temp = torch.full((codec.compress_temp_size(src.n, 65536),), 0xAB, dtype=torch.uint8, device=src.device)
    assert codec.compress_async(src, 65536, temp, dst) == 0
    torch.cuda.synchronize()
    tickets, counts, (repeats, looked, near) = _header(temp)
    if lz4_shape == "auto":                                             # (a forced shape skips the routing kernel)
        # (a chunk that the LDS shape's wave handed on to the sparse class -- lz4_common.hiph give_away: it opened
        # like data that compresses -- is on both lists; the chunks the far kernels gave back are counted in word 12)
        handed_on = sum(counts) - src.n
        given_back = temp[:64].view(torch.int32).cpu().tolist()[12]
        assert 0 <= handed_on <= counts[0] and 0 <= given_back <= sum(counts[1:])
        assert tickets[0] >= counts[0] and sum(tickets[1:]) >= sum(counts[1:])
        assert looked > 0 and repeats * 4 > looked                     # the samples' totals: compressible
        assert counts[0] >= 5 * 32                                     # random bytes, the empty chunk, the tiny one: LDS shape
        assert counts[1] + counts[2] + counts[3] >= 30 * 32            # text, the harness's data, runs: far shapes
    else:
        forced = {"mix": 0, "pair": 0, "far": 1, "fars": 2, "farw": 3}[lz4_shape]
        assert tickets[forced] >= src.n and sum(counts) == 0
    if lz4_shape in ("far", "fars", "farw"):                           # 1536 chunks: device-table waves beside the LDS ones
        assert bool((temp[256 + 16 * src.n:] != 0xAB).any().item())     # hash tables in the temp buffer were written
    got = dst.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % 48], f"chunk {i} {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, dst, chunks, dtype)
    compare_with_reference(reflib, "compressible batch", _reference_agrees(hc, base, dtype, 65536, want, tname))


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_several_sequences_per_trip_corner_cases(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """The far shapes' lean form takes several sequences off one trip to memory
    (far_straight_several): periodic data with periods shorter than its span,
    tiny alphabets (lanes of one window in one table slot, matches that overlap
    their source), vocabulary text with near and far candidates, real text.  64
    crafted chunks x 24, every chunk against the oracle and the reference
    build, then the round trip."""
    base = [c[: len(c) // es * es] for c in datagen.trip_corner_chunks()]
    chunks = base * 24
    want = [_want(oracle, c, es, 65536) for c in base]
    src, mine = _compress(hc, chunks, dtype, 65536)
    got = mine.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % len(base)], f"chunk {i} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "trip corner cases", _reference_agrees(hc, base, dtype, 65536, want, tname))


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_most_repetitive_batches_take_the_widest_span(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """When nearly every sampled word repeats (the harness's data, short periods,
    tiny alphabets) the far shape's lean form looks up 52 lanes per trip instead
    of 40: checked that the sampler says so, then every chunk against the oracle
    and the reference build, and the round trip."""
    import torch
    base = []
    for k in range(16):
        base.append(datagen.harness_like_int32(700 + k, 16384 - 5 * k).tobytes())
    for i, p in enumerate((1, 2, 3, 4, 5, 7, 8, 12, 16, 24)):
        base.append(datagen.periodic_bytes(720 + i, 65536 - 9 * i, p, 50 + 7 * i))
    for i, sy in enumerate((2, 2, 3, 3, 4, 4)):
        base.append(datagen.small_alphabet_bytes(740 + i, 65536 - 11 * i, sy))
    base = [c[: len(c) // es * es] for c in base]
    assert len(base) == 32
    chunks = base * 48
    want = [_want(oracle, c, es, 65536) for c in base]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype))
    dst = hc.batch.alloc_batch(src.n, codec.max_output_chunk_size(65536), src.device)
    temp = torch.zeros((codec.compress_temp_size(src.n, 65536),), dtype=torch.uint8, device=src.device)
    assert codec.compress_async(src, 65536, temp, dst) == 0
    torch.cuda.synchronize()
    tickets, counts, (repeats, looked, near) = _header(temp)
    if lz4_shape == "auto":
        assert looked > 0 and repeats * 8 > looked * 7
        assert counts[1] >= src.n * 3 // 4                              # the dense class: widest span, all 32 waves
    got = dst.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % len(base)], f"chunk {i} {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, dst, chunks, dtype)
    compare_with_reference(reflib, "most repetitive batch", _reference_agrees(hc, base, dtype, 65536, want, tname))


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_runs_of_element_sized_values(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """Run-length data whose values are of the element's size (the wide form's
    runs trip for 2- and 4-byte elements: all sequences of 48 lanes from one trip
    to the table), run lengths up to 2 .. 60 elements, ragged chunk lengths: every
    chunk against the oracle and the reference build, then the round trip."""
    base = []
    for k, longest in enumerate((2, 3, 4, 6, 8, 12, 16, 24, 32, 40, 60, 5)):
        for j in range(4):
            base.append(datagen.runs_of_elements(800 + 10 * k + j, 65536 - 4 * (k + 7 * j), es, longest))
    assert len(base) == 48
    chunks = base * 32
    want = [_want(oracle, c, es, 65536) for c in base]
    src, mine = _compress(hc, chunks, dtype, 65536)
    got = mine.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % len(base)], f"chunk {i} {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "runs of elements", _reference_agrees(hc, base, dtype, 65536, want, tname))


def test_small_tables_many_waves_per_group(hc, oracle, cuda, lz4_shape):
    """max_chunk below 16 KiB: smaller hash tables, up to 16 waves (= chunks
    in flight) per workgroup, batch sizes that do not fill the last group."""
    rng = np.random.default_rng(21)
    for max_chunk, nchunks in ((100, 37), (1000, 50), (4096, 33), (8192, 21)):
        chunks = []
        for i in range(nchunks):
            n = int(rng.integers(0, max_chunk + 1))
            kind = i % 3
            if kind == 0:
                chunks.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
            elif kind == 1:
                chunks.append(datagen.text_like(i, n))
            else:
                chunks.append(datagen.sparse_repeats(i, n, 150, 6) if n > 400 else bytes(rng.integers(0, 4, n, dtype=np.uint8)))
        src = hc.batch.from_host_chunks(chunks, "cuda:0")
        codec = hc.batch.Codec("LZ4", hc.LZ4Opts(0))
        mine = codec.compress(src, max_chunk)
        got = mine.to_host_chunks()
        for i, c in enumerate(chunks):
            assert got[i] == oracle.lz4_compress(c, 1, max_chunk), f"max_chunk={max_chunk} chunk {i}"
        dec, actual, statuses = codec.decompress(mine, max(max_chunk, 8))
        assert statuses.cpu().tolist() == [0] * len(chunks)
        assert dec.to_host_chunks() == chunks


def test_temp_buffer_of_exactly_the_contract_size(hc, oracle, cuda, lz4_shape):
    """The encoder keeps its chunk ticket counter in the temp buffer.  The
    contract size can be smaller than that counter (tiny max_chunk, tiny
    batch) and the caller's pointer need not be 4-byte aligned: nothing
    outside [temp, temp + temp_bytes) may be written, results unchanged."""
    import torch
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(0))
    for max_chunk, chunks in ((2, [b"ab"]), (3, [b"abc", b"x"]), (1, [b"z"]), (64, [bytes(range(64))] * 3), (65536, [datagen.text_like(7, 65536)] * 7)):
        src = hc.batch.from_host_chunks(chunks, "cuda:0")
        need = codec.compress_temp_size(src.n, max_chunk)
        for shift in (0, 1, 2, 3):
            arena = torch.full((need + 64,), 0xA5, dtype=torch.uint8, device="cuda:0")
            temp = arena[16 + shift: 16 + shift + need]
            dst = hc.batch.alloc_batch(src.n, codec.max_output_chunk_size(max(max_chunk, 1)), "cuda:0")
            assert codec.compress_async(src, max_chunk, temp, dst) == 0
            torch.cuda.synchronize()
            got = dst.to_host_chunks()
            for i, c in enumerate(chunks):
                assert got[i] == oracle.lz4_compress(c, 1, max_chunk), (max_chunk, shift, i)
            a = arena.cpu().numpy()
            assert (a[: 16 + shift] == 0xA5).all() and (a[16 + shift + need:] == 0xA5).all(), (max_chunk, shift)


def test_large_chunks_beyond_64k(hc, oracle, reflib, cuda, lz4_shape):
    """> 65536 elements: exercises the 16-bit position wrap of the hash table.
    In typed modes the reference truncates byte offsets > 65535 (a corrupt
    stream); the product rejects those candidates instead -- the one place
    where bytes differ on purpose, and only for chunks > 64 KiB."""
    rng = np.random.default_rng(5)
    base = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    chunks = [(base * 100)[:250000], datagen.text_like(3, 200001), bytes(rng.integers(0, 3, 150000, dtype=np.uint8))]
    faithful = {}
    for dtype, es in ((0, 1), (4, 4)):
        src, mine = _compress(hc, chunks, dtype, 250000)
        got = mine.to_host_chunks()
        faithful[es] = [_want(oracle, c, es, 250000, valid_offsets=False) for c in chunks]
        for i, c in enumerate(chunks):
            assert got[i] == _want(oracle, c, es, 250000, valid_offsets=True), (i, es, lz4_shape)
            if es == 1:
                assert got[i] == faithful[es][i]
        _round_trip(hc, mine, chunks, dtype, cap=250016)
    # the reference's own typed stream of the text chunk does not decode to the input
    st, out = oracle.lz4_decompress(faithful[4][1], 250016)
    assert out != chunks[1]

    def check(reflib):
        for dtype, es in ((0, 1), (4, 4)):
            _reference_agrees(hc, chunks, dtype, 250000, faithful[es], "reference-faithful oracle, elem %d" % es)(reflib)
    compare_with_reference(reflib, "chunks beyond 64 KiB", check)


def test_reference_harness_batches(hc, oracle, reflib, cuda, lz4_shape):
    """The six batches of tests/test_batch_c_api.h:772-777 with its data."""
    import torch
    mine_all = []
    for chunks in datagen.harness_batches():
        max_chunk = max(len(c) for c in chunks)
        for dtype, es in ((0, 1), (4, 4)):
            src, mine = _compress(hc, chunks, dtype, max_chunk)
            got = mine.to_host_chunks()
            mine_all.append((chunks, dtype, max_chunk, got))
            step = max(1, len(chunks) // 64)  # oracle on a sample (the reference build covers all)
            for i in range(0, len(chunks), step):
                assert got[i] == _want(oracle, chunks[i], es, max_chunk), (i, es, lz4_shape)
            codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype))
            # GetDecompressSize == input bytes (harness :366-384)
            assert codec.get_decompress_size(mine).cpu().tolist() == [len(c) for c in chunks]
            dec, actual, statuses = codec.decompress(mine, max_chunk)
            assert statuses.cpu().tolist() == [0] * len(chunks)       # harness :456-458
            assert actual.cpu().tolist() == [len(c) for c in chunks]
            assert dec.to_host_chunks() == chunks
            # nullptr actual_bytes / statuses accepted (harness :399-426)
            dec2, a2, s2 = codec.decompress(mine, max_chunk, with_status=False)
            torch.cuda.synchronize()
            dec2.sizes = src.sizes
            assert dec2.to_host_chunks() == chunks

    def check(reflib):
        for chunks, dtype, max_chunk, got in mine_all:
            _, ref = _compress(hc, chunks, dtype, max_chunk, lib=reflib)
            assert got == ref.to_host_chunks(), "kernel != reference build"
    compare_with_reference(reflib, "the harness's six batches (all chunks)", check)


def test_crash_safe_raw_input_is_rejected(hc, cuda):
    """Decompressing raw input as if it were compressed (harness :505-724):
    every chunk -> hipcompErrorCannotDecompress and size 0."""
    chunks = next(iter(b for b in datagen.harness_batches() if len(b) == 127))
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("LZ4")
    dec, actual, statuses = codec.decompress(src, max(len(c) for c in chunks))
    assert statuses.cpu().tolist() == [hc.hipcompStatus.ErrorCannotDecompress] * len(chunks)
    assert actual.cpu().tolist() == [0] * len(chunks)


def test_decoder_error_paths_match_oracle(hc, oracle, cuda):
    rng = np.random.default_rng(77)
    data = datagen.text_like(5, 30000)
    good = oracle.lz4_compress(data, 1, 65536)
    streams = [good, good[:-1], good[: len(good) // 2], good[:1], b"\xf0", b"\x10A\x00\x00", b"\x1fA\x01\x00",
               bytes(rng.integers(0, 256, 500, dtype=np.uint8)), b"", b"\x00"]
    for cap in (30000, 29999, 100):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        dec, actual, statuses = hc.batch.Codec("LZ4").decompress(comp, cap)
        st = statuses.cpu().tolist()
        ac = actual.cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.lz4_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes


def _corruptions(rng, good: bytes, n: int):
    """Valid stream with 1-3 bytes changed, a byte inserted or removed, or cut short."""
    out = []
    for k in range(n):
        b = bytearray(good)
        kind = k % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            del b[int(rng.integers(0, len(b)))]
        elif kind == 2:
            b.insert(int(rng.integers(0, len(b) + 1)), int(rng.integers(0, 256)))
        else:
            b = b[: int(rng.integers(1, len(b)))]
        out.append(bytes(b))
    return out


def test_corrupted_streams_decode_like_the_oracle(hc, oracle, cuda):
    """Status, reported size and (on success) bytes of damaged streams: the
    register-window fast path and the general path of the decoder must agree
    with the oracle everywhere, not only on valid input."""
    rng = np.random.default_rng(1234)
    sources = [datagen.text_like(11, 3000), datagen.harness_like_int32(12, 600).tobytes(),
               datagen.sparse_repeats(13, 4000, 90, 7), datagen.random_runs_int32(14, 700).tobytes()]
    streams = []
    for src in sources:
        good = oracle.lz4_compress(src, 1, 65536)
        streams += [good] + _corruptions(rng, good, 60)
    for cap in (4000, 2400):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        codec = hc.batch.Codec("LZ4")
        dec, actual, statuses = codec.decompress(comp, cap)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        sizes = codec.get_decompress_size(comp).cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.lz4_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes, (i, cap)
            sst, ssize = oracle.lz4_decompressed_size(s)
            assert sizes[i] == (ssize if sst == 0 else 0), (i, "size query")


def test_corrupted_chain_streams_decode_like_the_oracle(hc, oracle, cuda):
    """The decoder's loop with the CHAIN form (lz4_decode.hiph: chunks whose first sequences end in a run without
    literals -- the harness's kind of data -- take it for good): whole 64 KiB chunks of that kind, damaged in 240
    ways each, must decode like the oracle -- status, size, bytes -- and so must the size query."""
    rng = np.random.default_rng(4321)
    sources = [datagen.harness_like_int32(77, 16384).tobytes(), datagen.harness_like_int32(78, 9000).tobytes()]
    streams = []
    for src in sources:
        good = oracle.lz4_compress(src, 1, 65536)
        streams += [good] + _corruptions(rng, good, 240)
    for cap in (65536, 30000):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        codec = hc.batch.Codec("LZ4")
        dec, actual, statuses = codec.decompress(comp, cap)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        sizes = codec.get_decompress_size(comp).cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.lz4_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes, (i, cap)
            sst, ssize = oracle.lz4_decompressed_size(s)
            assert sizes[i] == (ssize if sst == 0 else 0), (i, "size query")


def test_decodes_liblz4_streams(hc, cuda):
    """Streams from the system liblz4 (a different, valid encoder) decode."""
    try:
        lz4 = ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("liblz4.so.1 not on this box")
    lz4.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lz4.LZ4_compressBound.argtypes = [ctypes.c_int]
    chunks = [c for _, c in datagen.edge_chunks() if len(c) > 0]
    comp = []
    for c in chunks:
        cap = lz4.LZ4_compressBound(len(c))
        buf = ctypes.create_string_buffer(cap)
        n = lz4.LZ4_compress_default(c, buf, len(c), cap)
        assert n > 0
        comp.append(buf.raw[:n])
    cb = hc.batch.from_host_chunks(comp, "cuda:0")
    codec = hc.batch.Codec("LZ4")
    dec, actual, statuses = codec.decompress(cb, 65536)
    assert statuses.cpu().tolist() == [0] * len(chunks)
    assert dec.to_host_chunks() == chunks
    assert codec.get_decompress_size(cb).cpu().tolist() == [len(c) for c in chunks]


def test_host_side_errors(hc, cuda):
    import torch
    lib = hc.default_library()
    src = hc.batch.from_host_chunks([b"abc" * 100], "cuda:0")
    codec = hc.batch.Codec("LZ4")
    dst = hc.batch.alloc_batch(1, codec.max_output_chunk_size(300), "cuda:0")
    small = torch.empty(8, dtype=torch.uint8, device="cuda:0")
    # temp too small -> InvalidValue (reference LZ4CompressionKernels.hip:173-180)
    assert codec.compress_async(src, 65536, small, dst) == hc.hipcompStatus.ErrorInvalidValue
    # LONGLONG unsupported (reference :217-218)
    big = torch.empty(32768, dtype=torch.uint8, device="cuda:0")
    bad = hc.batch.Codec("LZ4", hc.LZ4Opts(hc.hipcompType.LONGLONG))
    assert bad.compress_async(src, 65536, big, dst) == hc.hipcompStatus.ErrorInvalidValue
    # host pointer array -> InvalidValue (reference HipUtils.hip:91-108)
    hostptrs = (ctypes.c_void_p * 1)(src.data.data_ptr())
    st = lib.hipcompBatchedLZ4CompressAsync(ctypes.addressof(hostptrs), src.sizes.data_ptr(), 65536, 1,
                                            big.data_ptr(), big.numel(), dst.ptrs.data_ptr(), dst.sizes.data_ptr(),
                                            hc.LZ4Opts(0), None)
    assert st == hc.hipcompStatus.ErrorInvalidValue
    # chunk > 16 MiB
    out = ctypes.c_size_t(0)
    assert lib.hipcompBatchedLZ4CompressGetTempSize(1, (1 << 24) + 1, hc.LZ4Opts(0), ctypes.byref(out)) == 10
    assert lib.hipcompBatchedLZ4CompressGetMaxOutputChunkSize((1 << 24) + 1, hc.LZ4Opts(0), ctypes.byref(out)) == 10


def test_maximum_chunk_size_16MiB(hc, oracle, cuda):
    """The API's largest chunk (16 MiB): sizes, round trip, oracle equality on
    a compressible input (the oracle walks 16 Mi positions: keep it to one)."""
    import torch
    n = 1 << 24
    rng = np.random.default_rng(2)
    base = datagen.text_like(9, 1 << 16)
    data = (base * (n // len(base) + 1))[:n]
    src = hc.batch.from_host_chunks([data], "cuda:0")
    codec = hc.batch.Codec("LZ4")
    mine = codec.compress(src, n)
    torch.cuda.synchronize()
    got = mine.to_host_chunks()[0]
    assert got == oracle.lz4_compress(data, 1, n)
    dec, actual, statuses = codec.decompress(mine, n)
    assert statuses.cpu().tolist() == [0] and actual.cpu().tolist() == [n]
    assert dec.to_host_chunks()[0] == data


def test_cpu_interop_example(cuda):
    """examples/lz4_cpu_interop.c: GPU-compressed chunks decode with liblz4 block by block
    and as one LZ4 frame; liblz4-compressed chunks decompress on the GPU."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "lz4_cpu_interop")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for args in (["37", "65536"], ["300", "4000"]):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=120)
        if r.returncode == 77:
            pytest.skip("no liblz4 on this box")
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("OK") == 3


def test_small_far_classes_go_with_the_largest(hc, oracle, reflib, cuda, monkeypatch):
    """Routing: a far-type class with fewer than 2048 chunks that is not the largest gets no launch
    of its own (each launch ends with the tail of its slowest chunk); its chunks are worked off by
    the largest class's kernel.  3000 chunks of the harness's data with a few chunks of run-length
    data and of text among them: three lists, ONE ticket counter in use, every byte as the oracle's."""
    import torch
    monkeypatch.delenv("HIPCOMP_LZ4_SHAPE", raising=False)  # (the product library reads none anyway)
    base = [datagen.harness_like_int32(900 + k, 16384).tobytes() for k in range(8)]
    runs = [datagen.runs_of_elements(950 + k, 65536, 4, 12) for k in range(5)]
    text = [datagen.text_like(960 + k, 65536) for k in range(3)]
    chunks = [base[i % 8] for i in range(3000)]
    for j, c in enumerate(runs + text):
        chunks[137 + 311 * j] = c
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(0))
    dst = hc.batch.alloc_batch(src.n, codec.max_output_chunk_size(65536), src.device)
    temp = torch.zeros((codec.compress_temp_size(src.n, 65536),), dtype=torch.uint8, device=src.device)
    assert codec.compress_async(src, 65536, temp, dst) == 0
    torch.cuda.synchronize()
    tickets, counts, _ = _header(temp)
    assert sum(counts) == 3000 and counts[0] == 0
    assert counts[1] >= 2900 and counts[3] >= 3 and counts[1] + counts[2] + counts[3] == 3000
    assert tickets[1] >= 3000 and tickets[2] == 0 and tickets[3] == 0      # one launch took all three lists
    got = dst.to_host_chunks()
    want = {id(c): _want(oracle, c, 1, 65536) for c in base + runs + text}
    for i, c in enumerate(chunks):
        assert got[i] == want[id(c)], i
    _round_trip(hc, dst, chunks, 0)


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_lean_form_near_the_end_of_a_chunk(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """The far shapes' lean form works on windows that lie kFarLeanMargin elements or more in front of
    the chunk's end and leaves the rest to the one-window code: chunks of every length around that
    margin (and around the wide form's), of data whose sequences run right up to the end, so that
    the hand-over happens at every distance from the end -- each chunk against the oracle and the
    reference build, then the round trip."""
    rng = np.random.default_rng(77)
    chunks = []
    for k, n in enumerate(range(130 * es, 900 * es, 7 * es)):
        kind = k % 4
        if kind == 0:
            c = datagen.harness_like_int32(1000 + k, (n + 3) // 4 + 1).tobytes()[:n]
        elif kind == 1:
            c = datagen.text_like(1000 + k, n)
        elif kind == 2:
            c = datagen.periodic_bytes(1000 + k, n, 3 + k % 9, 40 + k % 17)
        else:
            c = datagen.runs_of_elements(1000 + k, n // es * es, es, 9)
        chunks.append(c[: len(c) // es * es])
    chunks = chunks * 12      # (more chunks than the LDS-table waves of a launch: device-table waves as well)
    base = chunks[: len(chunks) // 12]
    want = [_want(oracle, c, es, 65536) for c in base]
    src, mine = _compress(hc, chunks, dtype, 65536)
    got = mine.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % len(base)], f"chunk {i} ({len(chunks[i])} bytes) {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "chunks around the lean form's margin", _reference_agrees(hc, base, dtype, 65536, want, tname))


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_chains_of_sequences_without_literals(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """Data of few distinct values is one short match after the other, none with a literal; the dense
    class's kernel writes such a chain from a window's first lane on the short way (lz4_far.hiph,
    kFormChains), a match of 16 bytes or more behind it ends the trip.  Alphabets of 2 ... 16 values of
    1, 2, 4 and 8 bytes (few values: long matches right behind short ones, chains of one lane; many:
    chains broken by literals), chunk lengths that put the chain's end at every distance from the
    lean form's margin -- each chunk against the oracle and the reference build, then the round trip.
    (Round 3: a chain walk whose first v_readlane read a stale register wrote the chain's last
    sequence twice -- 25 times in the 64 KiB chunk of the edge set; check_asm_hazards.py H9.)"""
    chunks = []
    k = 0
    for values in (2, 3, 4, 5, 8, 16):
        for width in (1, 2, 4, 8):
            for n in (65536, 65536 - 36, 20000 + 4 * k, 3000 + k):
                rng = np.random.default_rng(9000 + k)
                alphabet = rng.integers(0, 256, (values, width), dtype=np.uint8)
                c = alphabet[rng.integers(0, values, n // width + 1)].reshape(-1)[:n].tobytes()
                chunks.append(c[: len(c) // es * es])
                k += 1
    base = chunks
    chunks = base * 16        # (1536 chunks: device-table waves beside the LDS-table ones)
    want = [_want(oracle, c, es, 65536) for c in base]
    src, mine = _compress(hc, chunks, dtype, 65536)
    got = mine.to_host_chunks()
    for i in range(len(chunks)):
        assert got[i] == want[i % len(base)], f"chunk {i} ({len(chunks[i])} bytes) {tname} shape={lz4_shape}: kernel != oracle"
    _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "chains of sequences without literals", _reference_agrees(hc, base, dtype, 65536, want, tname))


def _sequences(blk: bytes):
    """(literal bytes, match bytes) of every sequence of an LZ4 block"""
    i, out, n = 0, [], len(blk)
    while i < n:
        tok = blk[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = blk[i]; i += 1; lit += b
                if b != 255:
                    break
        i += lit
        if i >= n:
            out.append((lit, 0))
            break
        i += 2
        ml = (tok & 15) + 4
        if (tok & 15) == 15:
            while True:
                b = blk[i]; i += 1; ml += b
                if b != 255:
                    break
        out.append((lit, ml))
    return out


@pytest.mark.parametrize("tname,dtype,es", TYPES)
def test_match_at_lane_63_with_the_next_window_beyond_it(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    """The several-sequences trips of the far kernels at a span of 64 lanes (the dense class, and every
    wave with its table in LDS): a trip whose last sequence ends BEHIND lane 63 while lane 63 itself
    has a table match -- bit 63 of the trip's match mask set with `start` > 63 (pick(), lz4_far.hiph:
    nothing may be taken twice) -- and a chain of sequences without literals that runs through lane 63
    (the chain walk).  tests/datagen.py:lane63_case crafts both; that the oracle's stream has exactly
    the crafted sequences is asserted here, so the windows fall where the case wants them.  A few chunks
    (one wave each, its table in LDS: span 64 whatever the class) and many (all kinds of waves)."""
    cases = []
    for layout in (1, 2):
        found = 0
        for seed in range(40):
            data, want_seqs = datagen.lane63_case(es, layout, seed)
            z = _want(oracle, data, es, 65536)
            if _sequences(z)[1:1 + len(want_seqs)] == want_seqs:
                cases.append(data)
                found += 1
                if found == 3:
                    break
        assert found == 3, (es, layout)
    for chunks in (cases, cases * 700):
        src, mine = _compress(hc, chunks, dtype, 65536)
        got = mine.to_host_chunks()
        want = [_want(oracle, c, es, 65536) for c in chunks]
        for i in range(len(chunks)):
            assert got[i] == want[i], f"chunk {i} {tname} shape={lz4_shape}: kernel != oracle"
        _round_trip(hc, mine, chunks, dtype)
    compare_with_reference(reflib, "lane 63", _reference_agrees(hc, cases, dtype, 65536, [_want(oracle, c, es, 65536) for c in cases], tname))


def test_two_host_threads_on_two_streams(hc, cuda):
    """The library keeps no state of its own between or across calls (round 4's prefetch companion, with its
    side stream, events and lock, is gone): two host threads compressing on two streams at once must each get
    the bytes a lone call gets, and every call's arrays must be free to reuse once its own stream is
    synchronised.  (6000 chunks: the pair shape of the encoder, lz4_mix.hiph.)"""
    import threading
    import torch
    import bench
    n = 6000
    data = [bench.gen_data("uniform", 0, n, cuda, 0x5EED0100 + i) for i in range(2)]
    jobs = [bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d) for d in data]
    alone = []
    for job in jobs:
        job.compress()
        torch.cuda.synchronize()
        alone.append((job.comp.sizes.clone(), job.comp.data.clone()))
    streams = [torch.cuda.Stream(device=cuda) for _ in jobs]
    errors = []

    def run(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(6):
                    jobs[i].comp.data.zero_()
                    jobs[i].compress()
                jobs[i].decompress()
            streams[i].synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for job, (sizes, comp) in zip(jobs, alone):
        assert torch.equal(job.comp.sizes, sizes)
        stride = job.comp.stride
        A = job.comp.data[: n * stride].view(n, stride)
        B = comp[: n * stride].view(n, stride)
        idx = torch.arange(stride, device=cuda)[None, :] < sizes[:, None]
        assert bool(((A == B) | ~idx).all().item())
        job.verify()


def test_concurrent_decompress_calls_share_one_temp_buffer(hc, cuda):
    """The reference never touches the decompress temp buffer (src/lowlevel/LZ4CompressionKernels.hip:224-249), so its
    callers may hand ONE buffer to calls in flight on several streams.  Here a large call keeps its chunk ticket
    counter in one word of it -- a different word per call (lz4_kernels.hip, lz4_launch_decompress): two batches
    decompressed at once on two streams through the same temp buffer, several times over, must both come back whole
    (a shared counter would leave chunks undecoded or decode them twice)."""
    import torch
    import bench
    n = 12000   # (more chunks than the chip holds waves: the persistent grid with tickets)
    kinds = ("harness", "uniform")
    data = [bench.gen_data(k, 0, n, cuda, 0x5EED0200 + i) for i, k in enumerate(kinds)]
    jobs = [bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), d) for d in data]
    for job in jobs:
        job.compress()
    torch.cuda.synchronize()
    shared = jobs[0].dtemp
    jobs[1].dtemp = shared
    streams = [torch.cuda.Stream(device=cuda) for _ in jobs]
    for rep in range(4):
        for job in jobs:
            job.out.data.zero_()
            job.actual.zero_()
            job.statuses.fill_(-1)
        torch.cuda.synchronize()
        for job, st in zip(jobs, streams):
            with torch.cuda.stream(st):
                job.decompress()
        torch.cuda.synchronize()
        for job in jobs:
            job.verify()


def test_misjudged_chunks_are_handed_on(hc, oracle, cuda):
    """The routing kernel looks at 1 KiB of a chunk.  Where exactly that misrepresents the chunk (bench.gen_misrouted)
    the wave that meets it hands it on early -- the LDS shape to the sparse class, a far wave to the list that one
    more launch of the LDS shape works through (lz4_common.hiph give_away): the header's counts say so, and the
    bytes are the oracle's whoever ends up compressing a chunk."""
    import torch
    import bench
    n = 2048
    for kind, expect in (("text_random_samples", "to the sparse class"), ("random_text_samples", "back to the LDS shape"),
                         ("text_random_first", "to the sparse class")):
        data = bench.gen_misrouted(kind, n, cuda)
        src = hc.batch.from_device_buffer(data, 65536)
        codec = hc.batch.Codec("LZ4", hc.LZ4Opts(hc.hipcompType.CHAR))
        dst = hc.batch.alloc_batch(n, codec.max_output_chunk_size(65536), cuda)
        temp = torch.zeros(codec.compress_temp_size(n, 65536), dtype=torch.uint8, device=cuda)
        assert codec.compress_async(src, 65536, temp, dst) == 0
        torch.cuda.synchronize()
        header = temp[:64].view(torch.int32).cpu().tolist()
        counts, given_back = header[4:8], header[12]
        handed_on = sum(counts) - n
        if expect == "to the sparse class":
            assert counts[0] >= n * 9 // 10 and handed_on >= n * 9 // 10 and given_back == 0, (kind, counts, given_back)
        else:
            assert counts[0] <= n // 10 and given_back >= n * 9 // 10 and handed_on == 0, (kind, counts, given_back)
        host = data.view(n, 65536)[:4].cpu().numpy()
        got = dst.to_host_chunks()
        for i in range(4):
            assert got[i] == oracle.lz4_compress(host[i].tobytes(), 1, 65536), (kind, i)
        dec, actual, statuses = codec.decompress(dst, 65536)
        assert statuses.cpu().tolist() == [0] * n
        assert torch.equal(dec.data[: n * dec.stride].view(n, dec.stride)[:, :65536].reshape(-1), data)
