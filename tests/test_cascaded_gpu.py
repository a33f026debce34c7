"""Cascaded parity on the GPU through the C ABI: compressed bytes equal the CPU
oracle exactly (don't-care bytes are 0 in both) and equal the reference build
(oracle/_ref) under the don't-care mask; round trips for all eight integer
types and several option sets; the reference test's known answers and error
paths (reference tests/test_cascaded_batch.cpp)."""
import numpy as np
import pytest

import datagen
from conftest import compare_with_reference
from test_cascaded_oracle_cpu import NP, _predefined, _sorted_column, no_progress_streams

pytestmark = pytest.mark.gpu


def _inputs(t, oracle):
    rng = np.random.default_rng(100 + t)
    dt = NP[t]
    return [
        _sorted_column(5 + t, 16384).astype(dt).tobytes(),                       # 64 KiB (u32) sorted column
        np.repeat(rng.integers(0, 100, 300), rng.integers(1, 40, 300)).astype(dt).tobytes(),
        np.zeros(5000, dtype=dt).tobytes(),
        rng.integers(-100, 100, 3000).astype(dt).tobytes(),
        rng.integers(0, 2**31, 2000).astype(dt).tobytes(),                       # incompressible -> raw fallback
        np.arange(7, dtype=dt).tobytes(),
        np.array([42], dtype=dt).tobytes(),
        b"",
        _predefined([3, 9, 4, 0, 1], [1, 20, 13, 25, 6], t),
        _predefined([1, 2, 3, 4, 5, 6], [10, 6, 15, 1, 13, 9], t),
        datagen.harness_like_int32(t, 3000).astype(dt).tobytes(),
    ]


@pytest.mark.parametrize("opts", [(2, 1, 1), (2, 1, 0), (1, 0, 1), (0, 1, 1), (0, 0, 1), (1, 1, 0), (3, 2, 1), (0, 0, 0)])
def test_compress_parity_and_roundtrip(hc, oracle, reflib, cuda, opts):
    import torch
    R, D, bp = opts
    todo = []   # what the reference build is compared on, after the oracle asserts
    for t in range(8):
        chunks = _inputs(t, oracle)
        src = hc.batch.from_host_chunks(chunks, "cuda:0")
        copts = hc.CascadedOpts(4096, t, R, D, bp)
        codec = hc.batch.Codec("Cascaded", copts)
        mine = codec.compress(src)
        torch.cuda.synchronize()
        got = mine.to_host_chunks()
        wants = [oracle.cascaded_compress(c, t, R, D, bp) for c in chunks]
        for i, c in enumerate(chunks):
            assert got[i] == wants[i][0], f"type {t} opts {opts} input {i}: kernel != oracle"
        s = oracle.CASCADED_TYPE_SIZE[t]
        expect = [c[: len(c) // s * s] for c in chunks]
        assert codec.get_decompress_size(mine).cpu().tolist() == [len(e) for e in expect]
        dec, actual, statuses = codec.decompress(mine, 65536 * 2)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        for i, e in enumerate(expect):
            if len(chunks[i]) == 0:          # empty partition: 0 compressed bytes -> header too short
                assert st[i] == hc.hipcompStatus.ErrorCannotDecompress and ac[i] == 0
            else:
                assert st[i] == 0 and ac[i] == len(e), (t, opts, i)
                assert dec.chunk_bytes(i, ac[i]) == e
        todo.append((t, chunks, src, copts, mine, wants, expect))

    def check(reflib):
        for t, chunks, src, copts, mine, wants, expect in todo:
            r = hc.batch.Codec("Cascaded", copts, lib=reflib).compress(src)
            torch.cuda.synchronize()
            refgot = r.to_host_chunks()
            for i in range(len(chunks)):
                want, mask = wants[i]
                assert oracle.masked_equal(refgot[i], want, mask), f"type {t} opts {opts} input {i}: oracle != reference"
            # the reference decodes our streams (it dispatches on partition 0's type: same type everywhere here)
            rdec, ractual, rstat = hc.batch.Codec("Cascaded", copts, lib=reflib).decompress(mine, 65536 * 2)
            rs = rstat.cpu().tolist()
            for i, e in enumerate(expect):
                if len(chunks[i]):
                    assert rs[i] == 0, (t, opts, i)
                    assert rdec.chunk_bytes(i, len(e)) == e
    # the reference is undefined (endless loop) when a delta layer meets 0 elements
    # (CascadedKernels.hiph:323) and cannot decode D > R >= 1: compared for D <= 1 only
    if D <= 1:
        compare_with_reference(reflib, f"Cascaded opts {opts}", check)


def test_mixed_types_in_one_batch_decode_per_partition(hc, oracle, cuda):
    """The reference takes the element type from partition 0 only; here every
    partition is decoded with its own type byte."""
    streams, expect = [], []
    for t in (5, 1, 7, 2, 4, 0, 6, 3):
        data = _sorted_column(t, 3000).astype(NP[t]).tobytes()
        comp, _ = oracle.cascaded_compress(data, t, 2, 1, 1)
        streams.append(comp)
        expect.append(data)
    comp = hc.batch.from_host_chunks(streams, "cuda:0")
    dec, actual, statuses = hc.batch.Codec("Cascaded").decompress(comp, 32768)
    assert statuses.cpu().tolist() == [0] * len(streams)
    assert dec.to_host_chunks() == expect


def test_decoder_error_paths_match_oracle(hc, oracle, cuda):
    data = _sorted_column(9, 5000).tobytes()
    good, _ = oracle.cascaded_compress(data, 5, 2, 1, 1)
    raw, _ = oracle.cascaded_compress(data, 5, 0, 0, 0)
    bad_type = bytearray(good); bad_type[3] = 9
    bad_size = bytearray(good); bad_size[4:8] = (len(data) + 400).to_bytes(4, "little")
    streams = [good, good[:-8], good[:40], good[:4], b"", raw, raw[:-4], bytes(bad_type), bytes(bad_size),
               good[:8] + b"\x00" * 64] + no_progress_streams()
    for cap in (len(data), len(data) - 4, len(data) + 1000):
        comp = hc.batch.from_host_chunks(streams, "cuda:0")
        dec, actual, statuses = hc.batch.Codec("Cascaded").decompress(comp, cap)
        st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
        for i, s in enumerate(streams):
            ost, obytes = oracle.cascaded_decompress(s, cap)
            assert st[i] == ost, (i, cap)
            assert ac[i] == len(obytes), (i, cap)
            if ost == 0:
                assert dec.chunk_bytes(i, ac[i]) == obytes
    sizes = hc.batch.Codec("Cascaded").get_decompress_size(hc.batch.from_host_chunks(streams, "cuda:0")).cpu().tolist()
    assert sizes == [oracle.cascaded_decompressed_size(s) for s in streams]


@pytest.mark.parametrize("t", [5, 3, 7])
def test_hostile_array_lengths_do_not_wrap_the_bounds_test(hc, oracle, cuda, t):
    """A sub-chunk's array lengths come straight from the stream.  0xFFFFF000 and its like made the 32-bit sum
    `offset + roundUp4(length)` of the reference's bounds test (block_read :712-713) wrap and pass, and the
    decoder then read up to 4 KiB past the compressed buffer (round 4's advisor).  Every such stream is refused;
    the chunks beside it in the batch decode as ever."""
    dt = NP[t]
    data = _sorted_column(11, 4096).astype(dt).tobytes()
    good, _ = oracle.cascaded_compress(data, t, 2, 1, 1)
    assert good[0] == 2 and good[1] == 1 and good[2] == 1
    streams = [good]
    for word in (1, 2, 3):                      # rle_bytes of layer 0 and 1, final_bytes of the first sub-chunk
        for hostile in (0xFFFFF000, 0xFFFFFFFC, 0xFFFFFFFF, 0x80000000, 0xFFFFF000 + 8):
            bad = bytearray(good)
            at = 8 + 4 * word
            bad[at:at + 4] = hostile.to_bytes(4, "little")
            streams.append(bytes(bad))
    streams.append(good)
    comp = hc.batch.from_host_chunks(streams, "cuda:0")
    dec, actual, statuses = hc.batch.Codec("Cascaded").decompress(comp, len(data))
    st, ac = statuses.cpu().tolist(), actual.cpu().tolist()
    assert st[0] == 0 and st[-1] == 0 and dec.chunk_bytes(0, ac[0]) == data and dec.chunk_bytes(len(streams) - 1, ac[-1]) == data
    for i in range(1, len(streams) - 1):
        assert st[i] == hc.hipcompStatus.ErrorCannotDecompress and ac[i] == 0, (i, st[i], ac[i])


def test_host_side_errors(hc, cuda):
    import torch
    src = hc.batch.from_host_chunks([bytes(400)], "cuda:0")
    dst = hc.batch.alloc_batch(1, 408, "cuda:0")
    bad = hc.batch.Codec("Cascaded", hc.CascadedOpts(4096, 99, 2, 1, 1))
    assert bad.compress_async(src, 400, None, dst) == hc.hipcompStatus.ErrorInvalidValue
    toomany = hc.batch.Codec("Cascaded", hc.CascadedOpts(4096, 7, 14, 8, 1))
    assert toomany.compress_async(src, 400, None, dst) == hc.hipcompStatus.ErrorInvalidValue
    lib = hc.default_library()
    assert lib.hipcompBatchedCascadedDecompressAsync(None, None, None, None, 1, None, 0, None, None, None) == 10


def test_large_partitions(hc, oracle, reflib, cuda):
    """Partitions of many 4096-byte sub-chunks, incl. a ragged last one."""
    import torch
    chunks = [_sorted_column(3, 262144).tobytes(), _sorted_column(4, 100003).astype(np.uint16).tobytes() + b"\x07",
              np.repeat(np.arange(5000, dtype=np.int64), 37).tobytes()]
    todo = []
    for t, c in zip((5, 3, 6), chunks):
        src = hc.batch.from_host_chunks([c], "cuda:0")
        copts = hc.CascadedOpts(4096, t, 2, 1, 1)
        codec = hc.batch.Codec("Cascaded", copts)
        mine = codec.compress(src)
        torch.cuda.synchronize()
        want, mask = oracle.cascaded_compress(c, t, 2, 1, 1)
        assert mine.to_host_chunks()[0] == want
        s = oracle.CASCADED_TYPE_SIZE[t]
        dec, actual, statuses = codec.decompress(mine, len(c) + 16)
        assert statuses.cpu().tolist() == [0]
        assert dec.to_host_chunks()[0] == c[: len(c) // s * s]
        todo.append((src, copts, want, mask))

    def check(reflib):
        for src, copts, want, mask in todo:
            r = hc.batch.Codec("Cascaded", copts, lib=reflib).compress(src)
            torch.cuda.synchronize()
            assert oracle.masked_equal(r.to_host_chunks()[0], want, mask)
    compare_with_reference(reflib, "Cascaded large partitions", check)


def test_chunk_size_is_ignored_as_in_the_reference(hc, oracle, reflib, cuda):
    """The reference ignores opts.chunk_size (cascaded.h:93-100) and always writes 4096-byte
    sub-chunks; so does the product: whatever a caller passes, it gets the reference's bytes.  And a
    stream whose header claims something else in the high nibble of byte 2 (the extension of
    rounds 2-3, removed) is refused like any undecodable header."""
    data = _sorted_column(2, 16384).tobytes()
    src = hc.batch.from_host_chunks([data], "cuda:0")
    want, mask = oracle.cascaded_compress(data, 5, 2, 1, 1)
    for cb in (4096, 8192, 16384, 512, 0):
        assert hc.batch.Codec("Cascaded", hc.CascadedOpts(cb, 5, 2, 1, 1)).compress(src).to_host_chunks()[0] == want
    bad = [want[:2] + bytes([want[2] | (code << 4)]) + want[3:] for code in (1, 2, 7)]
    comp = hc.batch.from_host_chunks(bad + [want], "cuda:0")
    dec, actual, statuses = hc.batch.Codec("Cascaded").decompress(comp, 65536)
    assert statuses.cpu().tolist() == [12, 12, 12, 0] and actual.cpu().tolist() == [0, 0, 0, len(data)]
    assert dec.chunk_bytes(3, len(data)) == data
    for b in bad:
        assert oracle.cascaded_decompress(b, len(data)) == (12, b"")

    def check(reflib):
        for cb in (4096, 8192, 16384):
            got = hc.batch.Codec("Cascaded", hc.CascadedOpts(cb, 5, 2, 1, 1), lib=reflib).compress(src).to_host_chunks()[0]
            assert oracle.masked_equal(got, want, mask)
    compare_with_reference(reflib, "chunk_size ignored", check)


def test_option_selector_picks_what_measures_smallest(hc, cuda):
    """hipcomp/cascaded_select.h (an API of this library's own: the reference has no selector for the
    batched interface): the options it returns compress the WHOLE batch within 2 % of the best of its
    candidate sets, on columns with different characters -- sorted keys with repeats (RLE + delta),
    a low-cardinality dimension column (RLE alone), small random integers (bit-packing alone), and
    full-range random words (nothing helps: no compression)."""
    import torch
    rng = np.random.default_rng(11)
    n, per = 256, 16384
    columns = {
        "sorted": np.stack([np.cumsum(rng.integers(0, 4, per) * rng.integers(1, 9, per)) + rng.integers(0, 1 << 20)
                            for _ in range(n)]).astype(np.uint32),
        "dimension": np.stack([np.repeat(rng.integers(0, 50, per // 16), 16) for _ in range(n)]).astype(np.uint32),
        "small": rng.integers(0, 1000, (n, per)).astype(np.uint32),
        "random": rng.integers(0, 1 << 32, (n, per), dtype=np.uint64).astype(np.uint32),
    }
    lib = hc.default_library()
    candidates = [(0, 0, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1), (2, 0, 1), (0, 2, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1)]
    picked = {}
    for name, col in columns.items():
        src = hc.batch.from_host_chunks([row.tobytes() for row in col], "cuda:0")
        temp = torch.empty(lib.cascaded_select_temp_size(), dtype=torch.uint8, device="cuda:0")
        opts, ratio = lib.cascaded_select_opts(src.ptrs.data_ptr(), src.sizes.data_ptr(), src.n, hc.hipcompType.UINT,
                                               temp.data_ptr(), temp.numel(), torch.cuda.current_stream().cuda_stream)
        assert opts.chunk_size == 4096 and opts.type == hc.hipcompType.UINT
        sizes = {}
        for R, D, bp in candidates:
            comp = hc.batch.Codec("Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, R, D, bp)).compress(src)
            sizes[(R, D, bp)] = int(comp.sizes.sum().item())
        mine = (opts.num_RLEs, opts.num_deltas, opts.use_bp)
        picked[name] = mine
        assert mine in sizes, mine
        assert sizes[mine] <= 1.02 * min(sizes.values()), (name, mine, sizes)
        assert abs(ratio - src.sizes.sum().item() / sizes[mine]) / ratio < 0.25, (name, ratio)  # (a 16 KiB-per-partition sample)
        # and the stream decodes
        codec = hc.batch.Codec("Cascaded", hc.CascadedOpts(4096, hc.hipcompType.UINT, *mine))
        comp = codec.compress(src)
        dec, actual, statuses = codec.decompress(comp, per * 4)
        assert statuses.cpu().tolist() == [0] * n and dec.to_host_chunks() == [row.tobytes() for row in col]
    assert picked["random"] == (0, 0, 0) and picked["small"] == (0, 0, 1), picked
    assert picked["dimension"][0] >= 1 and picked["sorted"][1] >= 1, picked
    # an empty batch: the defaults
    opts, ratio = lib.cascaded_select_opts(0, 0, 0, hc.hipcompType.INT, 0, 0, 0)
    assert (opts.num_RLEs, opts.num_deltas, opts.use_bp, ratio) == (2, 1, 1, 1.0)


def test_random_partitions_of_every_bit_width(cuda):
    """scripts/parity_sweep_cascaded.py, two rounds: random partitions built to meet every bit width of the
    16-elements-per-lane packer / unpacker (values 1 .. 32 bits, run lengths 0 .. 10 bits, arrays ending in every
    position of a block), all eight types x eight option sets -- kernel bytes == oracle bytes, the round trip,
    and the kernel decoding the oracle's streams.  (profiles/r04_cascaded_parity_sweep.txt: forty rounds.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "parity_sweep_cascaded.py"), "2"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TOTAL BAD 0" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
