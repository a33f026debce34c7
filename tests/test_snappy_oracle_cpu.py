"""The Snappy CPU oracle against what the reference's own tests pin:
decoder golden vectors (tests/test_snappy_app.cpp:210-223) and the encoder /
decoder known answers of src/test/SnappyLargeTokens_test.cpp:381-537
(rebuilt here from the Snappy format description)."""
import json
import os

import numpy as np
import pytest

import datagen

HERE = os.path.dirname(os.path.abspath(__file__))


def varint(n):
    out = bytearray()
    while n > 0x7F:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def literal_tag(n):
    n -= 1
    if n < 60:
        return bytes([n << 2])
    nb = (n.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little")


def copy_tag(offset, length):
    if 4 <= length <= 11 and offset <= 2047:
        return bytes([((offset >> 8) << 5) | ((length - 4) << 2) | 1, offset & 0xFF])
    if offset <= 0xFFFF:
        return bytes([((length - 1) << 2) | 2]) + offset.to_bytes(2, "little")
    return bytes([((length - 1) << 2) | 3]) + offset.to_bytes(4, "little")


def test_reference_decoder_golden_vectors(oracle):
    with open(os.path.join(HERE, "golden", "snappy_app_vectors.json")) as f:
        vecs = json.load(f)["vectors"]
    assert len(vecs) == 2
    for v in vecs:
        comp, exp = bytes.fromhex(v["compressed_hex"]), bytes.fromhex(v["expected_hex"])
        assert oracle.snappy_uncompressed_size(comp) == len(exp)
        assert oracle.snappy_decompress(comp, len(exp)) == (0, exp), v["name"]
        assert oracle.snappy_decompress(comp, 0) == (0, exp)


def test_encoder_known_answers(oracle):
    # 256 distinct literals (SnappyLargeTokens_test.cpp:381-405)
    data = bytes(range(256))
    assert oracle.snappy_compress(data) == varint(256) + literal_tag(256) + data
    # 256 literals + copy(offset 256, length 64) (:410-447)
    data2 = bytes(range(256)) + bytes(range(64))
    assert oracle.snappy_compress(data2) == varint(320) + literal_tag(256) + bytes(range(256)) + copy_tag(256, 64)


def test_decoder_accepts_tokens_the_gpu_encoder_never_emits(oracle):
    # > 256-byte literal (:452-476)
    data = bytes(i % 256 for i in range(512))
    assert oracle.snappy_decompress(varint(512) + literal_tag(512) + data, 512) == (0, data)
    # 2-byte offset > 32 KiB and 4-byte offset (:481-537)
    rng = np.random.default_rng(42)
    for n in ((1 << 15) + 100, (1 << 16) + 100):
        vals = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        full = vals + vals[:35]
        comp = varint(len(full)) + literal_tag(n) + vals + copy_tag(n, 35)
        assert oracle.snappy_decompress(comp, len(full)) == (0, full)


def test_roundtrip_and_bounds(oracle):
    for name, data in datagen.edge_chunks():
        comp = oracle.snappy_compress(data)
        assert len(comp) <= oracle.snappy_max_compressed_size(len(data)), name
        assert oracle.snappy_uncompressed_size(comp) == len(data)
        assert oracle.snappy_decompress(comp, max(len(data), 1) if data else 0) == (0, data), name


def test_decoder_errors(oracle):
    data = datagen.text_like(4, 5000)
    good = oracle.snappy_compress(data)
    assert oracle.snappy_decompress(good, 4999)[0] == 12           # capacity < stream size
    st, out = oracle.snappy_decompress(good[:-5], 5000)            # truncated
    assert st == 12 and data.startswith(out)
    assert oracle.snappy_decompress(b"", 10) == (12, b"")
    assert oracle.snappy_decompress(varint(10) + copy_tag(5, 4), 10)[0] == 12   # copy before start
    assert oracle.snappy_decompress(b"\xff\xff\xff\xff\x7f", 10)[0] == 12        # size >= 2^31
    assert oracle.snappy_uncompressed_size(b"\xff\xff\xff\xff\x7f") == 0
    assert oracle.snappy_decompress(b"\x00", 0) == (0, b"")
    raw = datagen.harness_like_int32(3, 2000).tobytes()                           # harness CRASH_SAFE
    assert oracle.snappy_decompress(raw, len(raw))[0] == 12
