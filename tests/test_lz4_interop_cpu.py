"""The LZ4 frame helpers (hipcomp/lz4_interop.h) on the CPU, against the system liblz4's
frame API: blocks with the reference encoder's exact bytes (oracle) wrapped into a frame
decode with LZ4F_decompress, and a frame made by LZ4F_compressFrame is cut into blocks
that decode with the block decoder (oracle restatement of the reference's)."""
import ctypes
from ctypes import c_int, c_size_t, c_uint, c_uint64, c_void_p

import numpy as np
import pytest

import datagen


def _liblz4():
    try:
        return ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("no liblz4 on this box")


def _helpers(hc):
    L = ctypes.CDLL(hc.default_library().path)
    L.hipcompLZ4FrameBound.restype = c_size_t
    L.hipcompLZ4FrameBound.argtypes = [c_size_t, c_size_t]
    return L


def _lz4f_decompress(lz4, frame: bytes, cap: int) -> bytes:
    lz4.LZ4F_createDecompressionContext.restype = c_size_t
    lz4.LZ4F_decompress.restype = c_size_t
    ctx = c_void_p()
    assert not lz4.LZ4F_isError(c_size_t(lz4.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100)))
    out = ctypes.create_string_buffer(max(cap, 1))
    src = ctypes.create_string_buffer(frame, len(frame))
    i = o = 0
    while i < len(frame):
        d, s = c_size_t(cap - o), c_size_t(len(frame) - i)
        r = lz4.LZ4F_decompress(ctx, ctypes.byref(out, o), ctypes.byref(d), ctypes.byref(src, i), ctypes.byref(s), None)
        assert not lz4.LZ4F_isError(c_size_t(r))
        i += s.value
        o += d.value
        if r == 0:
            break
    lz4.LZ4F_freeDecompressionContext(ctx)
    return out.raw[:o]


def _frame_from_blocks(L, blocks, sizes):
    n = len(blocks)
    bufs = [ctypes.create_string_buffer(b, max(len(b), 1)) for b in blocks]
    ptrs = (c_void_p * max(n, 1))(*[ctypes.cast(b, c_void_p) for b in bufs])
    cb = (c_size_t * max(n, 1))(*[len(b) for b in blocks])
    ub = (c_size_t * max(n, 1))(*sizes)
    cap = L.hipcompLZ4FrameBound(n, sum(len(b) for b in blocks))
    frame = ctypes.create_string_buffer(cap)
    fb = c_size_t(0)
    st = L.hipcompLZ4FrameFromBlocks(ptrs, cb, ub, c_size_t(n), frame, c_size_t(cap), ctypes.byref(fb))
    return st, frame.raw[: fb.value]


def test_frame_of_reference_exact_blocks_decodes_with_liblz4(hc, oracle):
    lz4, L = _liblz4(), _helpers(hc)
    chunks = [c for _, c in datagen.edge_chunks() if len(c) <= 65536]
    for es in (1, 4):
        use = [c for c in chunks if len(c) % es == 0]
        blocks = [oracle.lz4_compress(c, es, 65536) for c in use]
        st, frame = _frame_from_blocks(L, blocks, [len(c) for c in use])
        assert st == 0
        assert frame[:4] == bytes([0x04, 0x22, 0x4D, 0x18])
        assert _lz4f_decompress(lz4, frame, sum(map(len, use)) + 16) == b"".join(use)
    # larger chunks pick a larger declared block size; above 4 MiB there is none
    big = bytes(np.random.default_rng(1).integers(0, 4, 300000, dtype=np.uint8))
    st, frame = _frame_from_blocks(L, [oracle.lz4_compress(big, 1, len(big))], [len(big)])
    assert st == 0 and (frame[5] >> 4) == 6
    assert _lz4f_decompress(lz4, frame, len(big) + 16) == big
    st, _ = _frame_from_blocks(L, [b"x"], [(4 << 20) + 1])
    assert st == 10
    st, frame = _frame_from_blocks(L, [], [])
    assert st == 0 and _lz4f_decompress(lz4, frame, 16) == b""


class _FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", c_int), ("blockMode", c_int), ("contentChecksumFlag", c_int), ("frameType", c_int),
                ("contentSize", c_uint64), ("dictID", c_uint), ("blockChecksumFlag", c_int)]


class _Prefs(ctypes.Structure):
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", c_int), ("autoFlush", c_uint),
                ("favorDecSpeed", c_uint), ("reserved", c_uint * 3)]


@pytest.mark.parametrize("block_checksums", [0, 1])
def test_liblz4_frame_is_cut_into_blocks(hc, oracle, block_checksums):
    lz4, L = _liblz4(), _helpers(hc)
    lz4.LZ4F_compressFrameBound.restype = c_size_t
    lz4.LZ4F_compressFrame.restype = c_size_t
    data = datagen.text_like(3, 300000) + bytes(np.random.default_rng(2).integers(0, 256, 70000, dtype=np.uint8))
    prefs = _Prefs()
    prefs.frameInfo.blockSizeID = 4          # 64 KiB
    prefs.frameInfo.blockMode = 1            # independent blocks
    prefs.frameInfo.contentChecksumFlag = 1
    prefs.frameInfo.blockChecksumFlag = block_checksums
    cap = lz4.LZ4F_compressFrameBound(c_size_t(len(data)), ctypes.byref(prefs))
    buf = ctypes.create_string_buffer(cap)
    n = lz4.LZ4F_compressFrame(buf, c_size_t(cap), data, c_size_t(len(data)), ctypes.byref(prefs))
    assert not lz4.LZ4F_isError(c_size_t(n))
    frame = buf.raw[:n]
    nb, bmax, content = c_size_t(0), c_size_t(0), c_uint64(0)
    assert L.hipcompLZ4FrameToBlocks(frame, c_size_t(n), ctypes.byref(nb), ctypes.byref(bmax), ctypes.byref(content),
                                     None, None, None, c_size_t(0)) == 0
    assert bmax.value == 65536 and nb.value == (len(data) + 65535) // 65536
    offs, sizes, raw = (c_size_t * nb.value)(), (c_size_t * nb.value)(), (c_int * nb.value)()
    assert L.hipcompLZ4FrameToBlocks(frame, c_size_t(n), ctypes.byref(nb), ctypes.byref(bmax), None, offs, sizes, raw,
                                     c_size_t(nb.value)) == 0
    out = b""
    for i in range(nb.value):
        blk = frame[offs[i]: offs[i] + sizes[i]]
        if raw[i]:
            out += blk                       # stored block (the random tail does not compress)
        else:
            st, dec = oracle.lz4_decompress(blk, 65536)
            assert st == 0
            out += dec
    assert out == data and any(raw) and not all(raw)
    # arrays too small / damaged frames
    assert L.hipcompLZ4FrameToBlocks(frame, c_size_t(n), ctypes.byref(nb), ctypes.byref(bmax), None, offs, sizes, raw,
                                     c_size_t(1)) == 10
    assert L.hipcompLZ4FrameToBlocks(frame, c_size_t(n - 9), ctypes.byref(nb), ctypes.byref(bmax), None, None, None, None,
                                     c_size_t(0)) == 12
    bad = bytearray(frame); bad[5] ^= 0x10
    assert L.hipcompLZ4FrameToBlocks(bytes(bad), c_size_t(n), ctypes.byref(nb), ctypes.byref(bmax), None, None, None, None,
                                     c_size_t(0)) == 12
    prefs.frameInfo.blockMode = 0            # linked blocks cannot be decoded as a batch
    n2 = lz4.LZ4F_compressFrame(buf, c_size_t(cap), data, c_size_t(len(data)), ctypes.byref(prefs))
    assert L.hipcompLZ4FrameToBlocks(buf.raw[:n2], c_size_t(n2), ctypes.byref(nb), ctypes.byref(bmax), None, None, None,
                                     None, c_size_t(0)) == 11
