"""Parity at odd addresses.  The C API allows ANY byte alignment for the chunks of LZ4 in its CHAR mode and of
Snappy -- inputs, compressed buffers and decode targets alike (typed LZ4 modes and Cascaded: element alignment) --
while every other test of this suite lays its batches out on 16-byte strides (hipcomp-core_amd/batch.py).  The kernels
are full of 16-byte loads and stores with alignment heads and tails; here every chunk of a batch sits at a byte
offset of 1, 2, 3, 5, 7, 9 or 15 from a 16-byte boundary, differently for input, compressed buffer and decode
target, and the bytes are compared with the oracle and with the reference build.  All launch shapes of the LZ4
encoder."""
import numpy as np
import pytest

import datagen
from conftest import compare_with_reference

pytestmark = pytest.mark.gpu

ODD = (1, 2, 3, 5, 7, 9, 15)


class Staggered:
    """n slots of `room` bytes, slot i beginning at 16-byte boundary + offsets[(i + turn) % len]"""

    def __init__(self, hc, n, room, device, offsets=ODD, turn=0, chunks=None):
        import torch
        self.slot = (room + 15) // 16 * 16 + 32
        self.offs = [offsets[(i + turn) % len(offsets)] for i in range(n)]
        host = np.full(n * self.slot + 64, 0xA5, dtype=np.uint8)
        sizes = [0] * n
        if chunks is not None:
            for i, c in enumerate(chunks):
                at = i * self.slot + self.offs[i]
                host[at:at + len(c)] = np.frombuffer(c, dtype=np.uint8)
                sizes[i] = len(c)
        self.data = torch.from_numpy(host).to(device)
        base = self.data.data_ptr()
        assert base % 16 == 0
        self.ptrs = torch.tensor([base + i * self.slot + self.offs[i] for i in range(n)], dtype=torch.int64, device=device)
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=device)
        self.batch = hc.batch.ChunkBatch(self.data, self.ptrs, self.sizes, self.slot)
        self.n = n

    def chunks(self, sizes=None):
        host = self.data.cpu().numpy()
        sizes = self.sizes.cpu().tolist() if sizes is None else sizes
        return [host[i * self.slot + self.offs[i]: i * self.slot + self.offs[i] + int(sizes[i])].tobytes() for i in range(self.n)]

    def untouched_outside(self, sizes):
        """nothing was written outside the chunks (the fill pattern is still there)"""
        host = self.data.cpu().numpy()
        keep = np.ones(host.size, dtype=bool)
        for i in range(self.n):
            at = i * self.slot + self.offs[i]
            keep[at:at + int(sizes[i])] = False
        return bool((host[keep] == 0xA5).all())


def _chunks(es):
    rng = np.random.default_rng(4242 + es)
    text = (b"the quick brown fox jumps over the lazy dog; " * 1600)[:65536]
    runs = np.repeat(rng.integers(0, 256, 9000, dtype=np.uint8), rng.integers(1, 17, 9000))[:65536].tobytes()
    out = [
        bytes(rng.integers(0, 256, 65536, dtype=np.uint8)),
        datagen.harness_like_int32(7, 16384).astype(np.int32).tobytes(),
        text, runs,
        bytes(rng.integers(0, 256, 32768 + 4 * es, dtype=np.uint8)),
        text[: 4096 + 4 * es], runs[: 300 * es], b"", bytes(rng.integers(0, 4, 13 * es, dtype=np.uint8)),
        bytes(rng.integers(0, 256, 65536 - 4 * es, dtype=np.uint8)), (b"ab" * 40000)[: 65536 - 8 * es],
        datagen.harness_like_int32(8, 5000).astype(np.int32).tobytes(),
    ]
    return [c[: len(c) // es * es] for c in out]


@pytest.mark.parametrize("tname,dtype,es", [("CHAR", 0, 1), ("USHORT", 3, 2), ("INT", 4, 4)])
def test_lz4_at_odd_addresses(hc, oracle, reflib, cuda, lz4_shape, tname, dtype, es):
    import torch
    from test_lz4_gpu import _want
    chunks = _chunks(es)
    # typed modes: the element's alignment is what the API asks for -- odd multiples of it
    offsets = ODD if es == 1 else tuple(sorted({(o * es) % 16 or es for o in ODD}))
    codec = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype))
    want = [_want(oracle, c, es, 65536) for c in chunks]

    def compress(lib_codec, turn):
        src = Staggered(hc, len(chunks), 65536, cuda, offsets, turn, chunks)
        out = Staggered(hc, len(chunks), lib_codec.max_output_chunk_size(65536), cuda, ODD, turn + 3)
        temp = torch.empty(max(lib_codec.compress_temp_size(len(chunks), 65536), 8), dtype=torch.uint8, device=cuda)
        assert lib_codec.compress_async(src.batch, 65536, temp, out.batch) == 0
        torch.cuda.synchronize()
        return out

    for turn in range(3):
        out = compress(codec, turn)
        sizes = out.sizes.cpu().tolist()
        got = out.chunks()
        for i in range(len(chunks)):
            assert got[i] == want[i], f"chunk {i} as {tname}, shape {lz4_shape}, turn {turn}: kernel != oracle"
        assert out.untouched_outside([codec.max_output_chunk_size(65536)] * len(chunks))   # (nothing outside the buffers)
        # decode, every chunk to an odd address of another turn; capacity = exactly the chunk's size
        dst = Staggered(hc, len(chunks), 65536, cuda, ODD, turn + 5)
        caps = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=cuda)
        actual = torch.full((len(chunks),), -1, dtype=torch.int64, device=cuda)
        statuses = torch.full((len(chunks),), -1, dtype=torch.int32, device=cuda)
        dtemp = torch.empty(max(codec.decompress_temp_size(len(chunks), 65536), 8), dtype=torch.uint8, device=cuda)
        assert codec.decompress_async(out.batch, caps, actual, dtemp, dst.batch, statuses) == 0
        torch.cuda.synchronize()
        assert statuses.cpu().tolist() == [0] * len(chunks)
        assert actual.cpu().tolist() == [len(c) for c in chunks]
        assert dst.chunks([len(c) for c in chunks]) == chunks
        assert dst.untouched_outside([len(c) for c in chunks])
        sz = torch.full((len(chunks),), -1, dtype=torch.int64, device=cuda)
        assert codec.get_decompress_size_async(out.batch, sz) == 0
        assert sz.cpu().tolist() == [len(c) for c in chunks]

    def check(ref):
        rc = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=ref)
        assert compress(rc, 1).chunks() == want
    compare_with_reference(reflib, "LZ4 at odd addresses", check)


def test_snappy_at_odd_addresses(hc, oracle, reflib, cuda):
    import torch
    chunks = _chunks(1)
    codec = hc.batch.Codec("Snappy")
    want = [oracle.snappy_compress(c) for c in chunks]

    def compress(lib_codec, turn):
        src = Staggered(hc, len(chunks), 65536, cuda, ODD, turn, chunks)
        out = Staggered(hc, len(chunks), lib_codec.max_output_chunk_size(65536), cuda, ODD, turn + 3)
        assert lib_codec.compress_async(src.batch, 65536, None, out.batch) == 0
        torch.cuda.synchronize()
        return out

    for turn in range(3):
        out = compress(codec, turn)
        assert out.chunks() == want, f"turn {turn}"
        assert out.untouched_outside([codec.max_output_chunk_size(65536)] * len(chunks))
        dst = Staggered(hc, len(chunks), 65536, cuda, ODD, turn + 5)
        caps = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=cuda)
        actual = torch.full((len(chunks),), -1, dtype=torch.int64, device=cuda)
        statuses = torch.full((len(chunks),), -1, dtype=torch.int32, device=cuda)
        assert codec.decompress_async(out.batch, caps, actual, None, dst.batch, statuses) == 0
        torch.cuda.synchronize()
        assert statuses.cpu().tolist() == [0] * len(chunks)
        assert dst.chunks([len(c) for c in chunks]) == chunks
        assert dst.untouched_outside([len(c) for c in chunks])

    def check(ref):
        assert compress(hc.batch.Codec("Snappy", lib=ref), 1).chunks() == want
    compare_with_reference(reflib, "Snappy at odd addresses", check)


@pytest.mark.parametrize("t", [1, 3, 5, 7])
def test_cascaded_at_element_alignment(hc, oracle, cuda, t):
    """Cascaded: buffers 4-byte and element aligned (cascaded.h) -- every such offset within 16 bytes."""
    import torch
    from test_cascaded_oracle_cpu import NP, _sorted_column
    dt = NP[t]
    es = np.dtype(dt).itemsize
    step = max(4, es)
    offsets = tuple(range(step, 16, step)) or (0,)
    rng = np.random.default_rng(77 + t)
    chunks = [_sorted_column(3 + t, 65536 // es).astype(dt).tobytes(),
              np.repeat(rng.integers(0, 100, 300), rng.integers(1, 40, 300)).astype(dt).tobytes(),
              rng.integers(0, 2**31, 2000).astype(dt).tobytes(), np.arange(7, dtype=dt).tobytes(), b"",
              rng.integers(-100, 100, 3000).astype(dt).tobytes()]
    opts = hc.CascadedOpts(4096, t, 2, 1, 1)
    codec = hc.batch.Codec("Cascaded", opts)
    want = [oracle.cascaded_compress(c, t, 2, 1, 1)[0] for c in chunks]
    for turn in range(len(offsets)):
        src = Staggered(hc, len(chunks), 65536, cuda, offsets, turn, chunks)
        out = Staggered(hc, len(chunks), codec.max_output_chunk_size(65536), cuda, offsets, turn + 1)
        out.data.zero_()
        assert codec.compress_async(src.batch, 65536, None, out.batch) == 0
        torch.cuda.synchronize()
        assert out.chunks() == want, f"type {t} turn {turn}"
        dst = Staggered(hc, len(chunks), 65536, cuda, offsets, turn + 2)
        caps = torch.full((len(chunks),), 65536, dtype=torch.int64, device=cuda)
        actual = torch.full((len(chunks),), -1, dtype=torch.int64, device=cuda)
        statuses = torch.full((len(chunks),), -1, dtype=torch.int32, device=cuda)
        assert codec.decompress_async(out.batch, caps, actual, None, dst.batch, statuses) == 0
        torch.cuda.synchronize()
        # (an empty partition compresses to nothing, and nothing does not decode: status 12 as in the reference)
        assert statuses.cpu().tolist() == [0 if c else hc.hipcompStatus.ErrorCannotDecompress for c in chunks]
        assert dst.chunks([len(c) for c in chunks]) == chunks
