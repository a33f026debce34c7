"""The Cascaded CPU oracle against the wire-layout known answers of the
reference's own test (tests/test_cascaded_batch.cpp:91-150, 213-379: two
predefined inputs, {RLE=2, Delta=1, bp=0}, all eight integer types), the raw
fallback header (:600-611), size-query behaviour (:950-990) and round trips."""
import struct

import numpy as np
import pytest

import datagen

NP = datagen.CASCADED_NP


def _ru(a, b):
    return (a + b - 1) // b * b


_predefined = datagen.predefined


@pytest.mark.parametrize("t", range(8))
def test_reference_known_answers_layout(oracle, t):
    s = oracle.CASCADED_TYPE_SIZE[t]
    cases = [
        (_predefined([3, 9, 4, 0, 1], [1, 20, 13, 25, 6], t), [1, 20, 13, 25, 6], [1, 1, 1, 1], [6, -5, -4, 1], 3),
        (_predefined([1, 2, 3, 4, 5, 6], [10, 6, 15, 1, 13, 9], t), [10, 6, 15, 1, 13, 9], [5], [1], 1),
    ]
    for data, runs0, runs1, output, delta_value in cases:
        comp, mask = oracle.cascaded_compress(data, t, 2, 1, 0)
        assert len(comp) % 4 == 0 and len(comp) % s == 0
        # partition header: 2 RLE, 1 delta, no bitpacking, the type (test :104-108)
        assert struct.unpack_from("<I", comp, 0)[0] == 2 + (1 << 8) + (0 << 16) + (t << 24)
        assert struct.unpack_from("<I", comp, 4)[0] == len(data)
        cs = _ru(8, s)
        sizes = struct.unpack_from("<4I", comp, cs)
        assert sizes[1] == len(runs0) * 2 and sizes[2] == len(runs1) * 2 and sizes[3] == len(runs1) * s
        dpos = _ru(cs + 16, s)
        assert np.frombuffer(comp, dtype=NP[t], count=1, offset=dpos)[0] == NP[t](delta_value)
        p = _ru(dpos + s, 4)
        assert list(np.frombuffer(comp, dtype=np.uint16, count=len(runs0), offset=p)) == runs0
        p = _ru(p + 2 * len(runs0), 4)
        assert list(np.frombuffer(comp, dtype=np.uint16, count=len(runs1), offset=p)) == runs1
        p = _ru(_ru(p + 2 * len(runs1), 4), s)
        want = np.array(output).astype(NP[t])
        assert list(np.frombuffer(comp, dtype=NP[t], count=len(output), offset=p)) == list(want)
        # all the bytes the reference test reads are "meaningful" under the mask
        assert mask[:8] == b"\xff" * 8
        st, dec = oracle.cascaded_decompress(comp, len(data))
        assert (st, dec) == (0, data)
        # and with bitpacking the round trip holds too (test :382-384)
        comp_bp, _ = oracle.cascaded_compress(data, t, 2, 1, 1)
        assert oracle.cascaded_decompress(comp_bp, len(data)) == (0, data)


def test_fallback_and_size_queries(oracle):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 2**32, 1000, dtype=np.uint32).tobytes()   # incompressible
    comp, mask = oracle.cascaded_compress(data, 5, 2, 1, 1)
    assert comp[:4] == bytes([0, 0, 0, 5]) and len(comp) == 8 + len(data)   # header type<<24 (test :600-611)
    assert comp[8:] == data
    assert oracle.cascaded_decompress(comp, len(data)) == (0, data)
    assert oracle.cascaded_decompressed_size(comp) == len(data)
    assert oracle.cascaded_decompressed_size(comp[:4]) == 0                 # (test :950-990)
    assert oracle.cascaded_decompress(comp[:4], 100) == (12, b"")
    assert oracle.cascaded_decompress(comp, len(data) - 4)[0] == 12          # output too small (:794-814)
    assert oracle.cascaded_decompress(comp[:-8], len(data))[0] == 12         # truncated input
    assert oracle.cascaded_compress(b"", 5, 2, 1, 1) == (b"", b"")
    assert oracle.cascaded_max_compressed_size(65536) == 65544
    # explicit no-compression options
    c0, _ = oracle.cascaded_compress(data, 5, 0, 0, 0)
    assert c0 == comp


_sorted_column = datagen.sorted_column


@pytest.mark.parametrize("opts", [(2, 1, 1), (2, 1, 0), (1, 0, 1), (0, 1, 1), (0, 0, 1), (1, 1, 0), (3, 2, 1), (2, 2, 1), (1, 2, 0)])
def test_roundtrip_options_and_types(oracle, opts):
    R, D, bp = opts
    rng = np.random.default_rng(11)
    for t in range(8):
        dt = NP[t]
        s = oracle.CASCADED_TYPE_SIZE[t]
        inputs = [
            _sorted_column(5, 16384).astype(dt).tobytes(),
            np.repeat(rng.integers(0, 100, 300), rng.integers(1, 40, 300)).astype(dt).tobytes(),
            np.zeros(5000, dtype=dt).tobytes(),                      # single-run sub-chunks -> empty layers
            rng.integers(-100, 100, 3000).astype(dt).tobytes(),
            np.arange(7, dtype=dt).tobytes(),
            np.array([42], dtype=dt).tobytes(),
        ]
        for data in inputs:
            comp, mask = oracle.cascaded_compress(data, t, R, D, bp)
            assert len(comp) <= oracle.cascaded_max_compressed_size(len(data))
            assert len(comp) % 4 == 0
            st, dec = oracle.cascaded_decompress(comp, len(data))
            assert (st, dec) == (0, data), (t, opts, len(data))
            assert oracle.cascaded_decompressed_size(comp) == len(data)
            # don't-care bytes really are don't-care for the decoder
            noisy = bytes(b if m else 0xA5 for b, m in zip(comp, mask))
            assert oracle.cascaded_decompress(noisy, len(data)) == (0, data)


def test_trailing_partial_element_is_dropped(oracle):
    data = np.arange(100, dtype=np.uint32).tobytes() + b"\x01\x02"      # 402 bytes, 100 elements
    comp, _ = oracle.cascaded_compress(data, 5, 2, 1, 1)
    assert oracle.cascaded_decompressed_size(comp) == 400
    assert oracle.cascaded_decompress(comp, 402) == (0, data[:400])


def no_progress_streams():
    """Corrupt partitions whose sub-chunk size field would leave the decoder's
    cursor where it is (1..3), or wrap it backwards (huge), while the
    sub-chunk itself decodes to zero elements (endless loop)."""
    out = []
    for csz in (1, 2, 3, 0xFFFFFFF0, 0x80000000):
        # R=0, D=0, bp=1, UINT; sub-chunk = [csz][final_bytes] + bit-packed array of 0 elements
        out.append(struct.pack("<4BI", 0, 0, 1, 5, 16) + struct.pack("<II", csz, 8) + struct.pack("<II", 0, 0))
        # R=1, D=0, bp=0: one run of length 0 -> zero elements
        out.append(struct.pack("<4BI", 1, 0, 0, 5, 4) + struct.pack("<III", csz, 2, 4)
                   + struct.pack("<HH", 0, 0) + struct.pack("<I", 7))
    return out


@pytest.mark.timeout(60)
def test_sub_chunk_without_progress_is_rejected(oracle):
    for s in no_progress_streams():
        assert oracle.cascaded_decompress(s, 4096) == (12, b"")
    # the same layouts with an honest size field still decode (to nothing / fail on the count, not hang)
    ok = struct.pack("<4BI", 0, 0, 1, 5, 0) + struct.pack("<II", 16, 8) + struct.pack("<II", 0, 0)
    assert oracle.cascaded_decompress(ok, 4096) == (0, b"")


def test_a_stream_that_claims_larger_sub_chunks_is_rejected(oracle):
    """Byte 2 of the partition header is use_bp, 0 or 1.  (Rounds 2-3 of this library marked sub-chunks of
    8192 / 16384 bytes in its high nibble, an opt-in extension that was measured useless and removed:
    such a stream is refused, not mis-decoded.)"""
    data = _sorted_column(5, 16384).tobytes()
    good, _ = oracle.cascaded_compress(data, 5, 2, 1, 1)
    assert good[2] == 1 and oracle.cascaded_decompress(good, len(data)) == (0, data)
    for code in (1, 2, 7):
        bad = good[:2] + bytes([good[2] | (code << 4)]) + good[3:]
        assert oracle.cascaded_decompress(bad, len(data)) == (12, b"")
