"""Snappy parity on the GPU through the C ABI: compressed bytes vs the CPU
oracle and vs the reference build (oracle/_ref), round trips, the reference's
golden decoder vectors and large-token cases, error paths."""
import json
import os

import numpy as np
import pytest

import datagen
from conftest import compare_with_reference
from test_snappy_oracle_cpu import copy_tag, literal_tag, varint

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _chunks():
    named = datagen.edge_chunks()
    named += [("lt4_%d" % n, bytes(range(n))) for n in (1, 2, 3, 4, 5)]
    named += [("seq256", bytes(range(256))), ("seq256_copy64", bytes(range(256)) + bytes(range(64)))]
    named += [("csv", datagen.text_like(77, 65536).replace(b" ", b"|"))]
    return named


def test_compress_bit_exact_and_roundtrip(hc, oracle, reflib, cuda):
    import torch
    named = _chunks()
    chunks = [c for _, c in named]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("Snappy")
    mine = codec.compress(src)
    torch.cuda.synchronize()
    got = mine.to_host_chunks()
    want = [oracle.snappy_compress(c) for c in chunks]
    for i, (name, c) in enumerate(named):
        assert got[i] == want[i], f"{name}: kernel != oracle"
    assert codec.get_decompress_size(mine).cpu().tolist() == [len(c) for c in chunks]
    dec, actual, statuses = codec.decompress(mine, 65536)
    assert statuses.cpu().tolist() == [0] * len(chunks)
    assert actual.cpu().tolist() == [len(c) for c in chunks]
    assert dec.to_host_chunks() == chunks

    def check(reflib):
        r = hc.batch.Codec("Snappy", lib=reflib).compress(src)
        torch.cuda.synchronize()
        refgot = r.to_host_chunks()
        for i, (name, c) in enumerate(named):
            assert refgot[i] == want[i], f"{name}: oracle != reference build"
        # the reference decodes our streams too
        rdec, ractual, rstat = hc.batch.Codec("Snappy", lib=reflib).decompress(mine, 65536)
        assert rstat.cpu().tolist() == [0] * len(chunks)
        assert rdec.to_host_chunks() == chunks
    compare_with_reference(reflib, "Snappy edge chunks", check)


def test_several_elements_per_trip_corner_cases(hc, oracle, reflib, cuda):
    """The encoder's straight path takes several elements off one trip to memory:
    periodic data with periods shorter than its span, tiny alphabets (lanes of one
    window with one hash, copies that overlap their source), vocabulary text, real
    text -- every chunk against the oracle and the reference build, then the round
    trip."""
    import torch
    chunks = datagen.trip_corner_chunks()
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("Snappy")
    mine = codec.compress(src)
    torch.cuda.synchronize()
    got = mine.to_host_chunks()
    want = [oracle.snappy_compress(c) for c in chunks]
    for i in range(len(chunks)):
        assert got[i] == want[i], f"chunk {i}: kernel != oracle"
    dec, actual, statuses = codec.decompress(mine, 65536)
    assert statuses.cpu().tolist() == [0] * len(chunks)
    assert dec.to_host_chunks() == chunks

    def check(reflib):
        r = hc.batch.Codec("Snappy", lib=reflib).compress(src)
        torch.cuda.synchronize()
        refgot = r.to_host_chunks()
        for i in range(len(chunks)):
            assert refgot[i] == want[i], f"chunk {i}: oracle != reference build"
    compare_with_reference(reflib, "Snappy trip corner cases", check)


def test_reference_harness_batches(hc, oracle, reflib, cuda):
    import torch
    mine_all = []
    for bi, chunks in enumerate(datagen.harness_batches()):
        src = hc.batch.from_host_chunks(chunks, "cuda:0")
        codec = hc.batch.Codec("Snappy")
        mine = codec.compress(src)
        torch.cuda.synchronize()
        got = mine.to_host_chunks()
        mine_all.append((bi, src, got))
        step = max(1, len(chunks) // 64)
        for i in range(0, len(chunks), step):
            assert got[i] == oracle.snappy_compress(chunks[i])
        assert codec.get_decompress_size(mine).cpu().tolist() == [len(c) for c in chunks]
        dec, actual, statuses = codec.decompress(mine, max(len(c) for c in chunks))
        assert statuses.cpu().tolist() == [0] * len(chunks)
        assert dec.to_host_chunks() == chunks
        dec2, _, _ = codec.decompress(mine, max(len(c) for c in chunks), with_status=False)
        torch.cuda.synchronize()
        dec2.sizes = src.sizes
        assert dec2.to_host_chunks() == chunks

    def check(reflib):
        for bi, src, got in mine_all:
            r = hc.batch.Codec("Snappy", lib=reflib).compress(src)
            torch.cuda.synchronize()
            assert got == r.to_host_chunks(), f"batch {bi}: kernel != reference build"
    compare_with_reference(reflib, "Snappy, the harness's batches (all chunks)", check)


def test_reference_decoder_vectors_and_large_tokens(hc, cuda):
    with open(os.path.join(HERE, "golden", "snappy_app_vectors.json")) as f:
        vecs = json.load(f)["vectors"]
    streams = [bytes.fromhex(v["compressed_hex"]) for v in vecs]
    expect = [bytes.fromhex(v["expected_hex"]) for v in vecs]
    # tokens the GPU encoder never emits (SnappyLargeTokens_test.cpp:452-537)
    data = bytes(i % 256 for i in range(512))
    streams.append(varint(512) + literal_tag(512) + data)
    expect.append(data)
    rng = np.random.default_rng(42)
    for n in ((1 << 15) + 100, (1 << 16) + 100):
        vals = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        streams.append(varint(n + 35) + literal_tag(n) + vals + copy_tag(n, 35))
        expect.append(vals + vals[:35])
    comp = hc.batch.from_host_chunks(streams, "cuda:0")
    codec = hc.batch.Codec("Snappy")
    assert codec.get_decompress_size(comp).cpu().tolist() == [len(e) for e in expect]
    dec, actual, statuses = codec.decompress(comp, max(len(e) for e in expect))
    assert statuses.cpu().tolist() == [0] * len(streams)
    assert dec.to_host_chunks() == expect


def test_decoder_error_paths_match_oracle(hc, oracle, cuda):
    data = datagen.text_like(4, 5000)
    good = oracle.snappy_compress(data)
    raw = datagen.harness_like_int32(3, 2000).tobytes()
    streams = [good, good[:-5], good[:40], b"", varint(10) + copy_tag(5, 4), b"\xff\xff\xff\xff\x7f", b"\x00", raw,
               varint(100) + literal_tag(50) + b"x" * 10]
    for cap in (5000, 4999, 8000):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        dec, actual, statuses = hc.batch.Codec("Snappy").decompress(comp, cap)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.snappy_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes
    sizes = hc.batch.Codec("Snappy").get_decompress_size(hc.batch.from_host_chunks(streams, "cuda:0")).cpu().tolist()
    assert sizes == [oracle.snappy_uncompressed_size(s) for s in streams]


def test_corrupted_streams_decode_like_the_oracle(hc, oracle, cuda):
    """Status and reported size (and bytes on success) of damaged streams."""
    rng = np.random.default_rng(4321)
    sources = [datagen.text_like(21, 3000), datagen.harness_like_int32(22, 600).tobytes(),
               datagen.tpch_lineitem_text(23, 4000)]
    streams = []
    for src in sources:
        good = oracle.snappy_compress(src)
        streams.append(good)
        for k in range(60):
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
            streams.append(bytes(b))
    for cap in (4000, 2500):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        dec, actual, statuses = hc.batch.Codec("Snappy").decompress(comp, cap)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.snappy_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes, (i, cap)


def test_large_chunks(hc, oracle, reflib, cuda):
    """Chunks far beyond 64 KiB: the 16-bit hash-map positions wrap and copy
    distances stay <= 32768."""
    import torch
    rng = np.random.default_rng(8)
    chunks = [datagen.tpch_lineitem_text(5, 300000), datagen.text_like(6, 1 << 20),
              bytes(rng.integers(0, 4, 200001, dtype=np.uint8)), bytes(rng.integers(0, 256, 150000, dtype=np.uint8))]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    codec = hc.batch.Codec("Snappy")
    mine = codec.compress(src)
    torch.cuda.synchronize()
    got = mine.to_host_chunks()
    for i, c in enumerate(chunks):
        assert got[i] == oracle.snappy_compress(c), i
    dec, actual, statuses = codec.decompress(mine, 1 << 20)
    assert statuses.cpu().tolist() == [0] * len(chunks)
    assert dec.to_host_chunks() == chunks

    def check(reflib):
        r = hc.batch.Codec("Snappy", lib=reflib).compress(src)
        torch.cuda.synchronize()
        assert r.to_host_chunks() == got
    compare_with_reference(reflib, "Snappy large chunks", check)
