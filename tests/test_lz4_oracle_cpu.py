"""The LZ4 CPU oracle against independent checks available without a GPU:
its streams are valid LZ4 blocks (system liblz4 decodes them), its decoder
inverts its encoder, and the quirks SURVEY.md section 0 lists are present."""
import ctypes

import numpy as np
import pytest

import datagen


def _liblz4():
    try:
        lz4 = ctypes.CDLL("liblz4.so.1")
    except OSError:
        return None
    lz4.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lz4.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lz4.LZ4_compressBound.argtypes = [ctypes.c_int]
    return lz4


@pytest.mark.parametrize("es", [1, 2, 4])
def test_roundtrip_and_liblz4_decodes(oracle, es):
    lz4 = _liblz4()
    for name, data in datagen.edge_chunks():
        for max_chunk in (65536, len(data), 0):
            comp = oracle.lz4_compress(data, es, max_chunk)
            assert len(comp) <= oracle.lz4_max_compressed_size(len(data))
            st, dec = oracle.lz4_decompress(comp, len(data))
            assert (st, dec) == (0, data), name
            assert oracle.lz4_decompressed_size(comp) == (0, len(data))
            if lz4 is not None and data:
                buf = ctypes.create_string_buffer(len(data))
                assert lz4.LZ4_decompress_safe(comp, buf, len(comp), len(data)) == len(data), name
                assert buf.raw == data


def test_last_token_low_nibble_is_0xC(oracle):
    # SURVEY.md finding 4: uint8_t(0 - 4) & 0x0f in the final sequence
    comp = oracle.lz4_compress(b"abcdefgh", 1, 65536)
    assert comp[0] == (8 << 4) | 0xC and comp[1:] == b"abcdefgh"
    comp = oracle.lz4_compress(bytes(range(40)), 1, 65536)
    assert comp[0] == 0xFC and comp[1] == 40 - 15


def test_store_winner_changes_the_table_not_validity(oracle):
    data = datagen.text_like(21, 65536)
    a = oracle.lz4_compress(data, 1, 65536, store_winner=1)
    b = oracle.lz4_compress(data, 1, 65536, store_winner=0)
    for c in (a, b):
        assert oracle.lz4_decompress(c, len(data)) == (0, data)


def test_decoder_rejects_what_the_harness_feeds_it(oracle):
    # raw harness data parsed as a stream is corrupt (harness CRASH_SAFE)
    raw = datagen.harness_like_int32(3, 4000).tobytes()
    st, out = oracle.lz4_decompress(raw, len(raw))
    assert st == 12 and out == b""
    good = oracle.lz4_compress(raw, 1, 65536)
    assert oracle.lz4_decompress(good, len(raw) - 1)[0] == 12      # capacity too small
    assert oracle.lz4_decompress(good[:-3], len(raw))[0] == 12     # truncated stream
    assert oracle.lz4_decompress(b"\x10A\x00\x00", 100)[0] == 12   # offset 0


def test_decoder_accepts_liblz4_streams(oracle):
    lz4 = _liblz4()
    if lz4 is None:
        pytest.skip("no liblz4")
    for name, data in datagen.edge_chunks():
        if not data:
            continue
        cap = lz4.LZ4_compressBound(len(data))
        buf = ctypes.create_string_buffer(cap)
        n = lz4.LZ4_compress_default(data, buf, len(data), cap)
        assert oracle.lz4_decompress(buf.raw[:n], len(data)) == (0, data), name
