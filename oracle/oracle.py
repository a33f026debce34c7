"""ctypes wrapper over oracle/liboracle.so (the C restatements).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libhipcomp_ref.so")

# Which lane's value survives when several lanes of one wave execute ONE
# global_store_short to the same address (0 = lowest lane, 1 = highest lane,
# 2 = the order measured on MI355X by tests/test_hw_probes.py: last write in
# the sequence "for g in 0..3, for p in 3..0, for q in 0..3: lane 16g+4q+p").
STORE_WINNER_GFX950 = 2


def gfx950_store_order_key(lane: int):
    """Sort key: the lane with the largest key survives (measured, gfx950)."""
    return (lane >> 4, 3 - (lane & 3), (lane >> 2) & 3)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_lz4_hash_table_size.argtypes = [c_size_t]
        L.oracle_lz4_hash_table_size.restype = c_size_t
        L.oracle_lz4_max_compressed_size.argtypes = [c_size_t]
        L.oracle_lz4_max_compressed_size.restype = c_size_t
        L.oracle_lz4_compress_temp_size.argtypes = [c_size_t, c_size_t]
        L.oracle_lz4_compress_temp_size.restype = c_size_t
        L.oracle_lz4_decompress_temp_size.argtypes = [c_size_t]
        L.oracle_lz4_decompress_temp_size.restype = c_size_t
        L.oracle_lz4_compress_ex.argtypes = [c_char_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p,
                                             POINTER(c_size_t)]
        L.oracle_lz4_compress_ex.restype = c_int
        L.oracle_lz4_decompress.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]
        L.oracle_lz4_decompress.restype = c_int
        L.oracle_snappy_max_compressed_size.argtypes = [c_size_t]
        L.oracle_snappy_max_compressed_size.restype = c_size_t
        L.oracle_snappy_compress.argtypes = [c_char_p, c_size_t, c_void_p, POINTER(c_size_t)]
        L.oracle_snappy_compress.restype = c_int
        L.oracle_snappy_uncompressed_size.argtypes = [c_char_p, c_size_t]
        L.oracle_snappy_uncompressed_size.restype = c_size_t
        L.oracle_snappy_decompress.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]
        L.oracle_snappy_decompress.restype = c_int
        L.oracle_cascaded_max_compressed_size.argtypes = [c_size_t]
        L.oracle_cascaded_max_compressed_size.restype = c_size_t
        L.oracle_cascaded_compress.argtypes = [c_char_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, POINTER(c_size_t)]
        L.oracle_cascaded_compress.restype = c_int
        L.oracle_cascaded_decompressed_size.argtypes = [c_char_p, c_size_t]
        L.oracle_cascaded_decompressed_size.restype = c_size_t
        L.oracle_cascaded_decompress.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]
        L.oracle_cascaded_decompress.restype = c_int
        _lib = L
    return _lib


# ---- LZ4 -------------------------------------------------------------------

def lz4_max_compressed_size(n: int) -> int:
    return lib().oracle_lz4_max_compressed_size(n)


def lz4_hash_table_size(max_chunk: int) -> int:
    return lib().oracle_lz4_hash_table_size(max_chunk)


def lz4_compress(data: bytes, elem_size: int = 1, max_chunk_bytes: int | None = None,
                 store_winner: int = STORE_WINNER_GFX950, valid_offsets: bool = True) -> bytes:
    """valid_offsets=True is the product's behaviour; False restates the
    reference exactly (they differ only for typed chunks > 64 KiB, where the
    reference emits truncated offsets)."""
    if max_chunk_bytes is None:
        max_chunk_bytes = len(data)
    cap = lz4_max_compressed_size(len(data)) + 16
    out = ctypes.create_string_buffer(cap)
    n = c_size_t(0)
    rc = lib().oracle_lz4_compress_ex(data, len(data), elem_size, max_chunk_bytes, store_winner,
                                      1 if valid_offsets else 0, ctypes.cast(out, c_void_p), ctypes.byref(n))
    if rc != 0:
        raise ValueError("oracle_lz4_compress: bad arguments")
    return out.raw[: n.value]


def lz4_decompress(comp: bytes, capacity: int):
    """-> (status, bytes).  status 0 / 12 as the reference reports it."""
    out = ctypes.create_string_buffer(max(capacity, 1))
    n = c_size_t(0)
    st = lib().oracle_lz4_decompress(comp, len(comp), ctypes.cast(out, c_void_p), capacity, ctypes.byref(n))
    return st, out.raw[: n.value]


def lz4_decompressed_size(comp: bytes):
    """Size-only pass (reference lz4BatchGetDecompressSizes)."""
    n = c_size_t(0)
    st = lib().oracle_lz4_decompress(comp, len(comp), None, 0, ctypes.byref(n))
    return st, n.value


# ---- Snappy ----------------------------------------------------------------

def snappy_max_compressed_size(n: int) -> int:
    return lib().oracle_snappy_max_compressed_size(n)


def snappy_compress(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(snappy_max_compressed_size(len(data)) + 16)
    n = c_size_t(0)
    lib().oracle_snappy_compress(data, len(data), ctypes.cast(out, c_void_p), ctypes.byref(n))
    return out.raw[: n.value]


def snappy_uncompressed_size(comp: bytes) -> int:
    return lib().oracle_snappy_uncompressed_size(comp, len(comp))


def snappy_decompress(comp: bytes, capacity: int):
    """-> (status, bytes produced).  capacity 0 = the stream's own size."""
    room = max(capacity, snappy_uncompressed_size(comp) if capacity == 0 else capacity, 1)
    out = ctypes.create_string_buffer(room)
    n = c_size_t(0)
    st = lib().oracle_snappy_decompress(comp, len(comp), ctypes.cast(out, c_void_p), capacity, ctypes.byref(n))
    return st, out.raw[: n.value]


# ---- Cascaded --------------------------------------------------------------

CASCADED_TYPE_SIZE = {0: 1, 1: 1, 2: 2, 3: 2, 4: 4, 5: 4, 6: 8, 7: 8}


def cascaded_max_compressed_size(n: int) -> int:
    return lib().oracle_cascaded_max_compressed_size(n)


def cascaded_compress(data: bytes, type_tag: int, num_rles: int, num_deltas: int, use_bp: int):
    """-> (compressed bytes, mask bytes).  mask 0xFF = byte defined by the
    format, 0x00 = don't-care byte (stale LDS / unwritten gap in the reference).
    (Sub-chunks are the reference's 4096 bytes: it ignores opts.chunk_size.)"""
    cap = cascaded_max_compressed_size(len(data))
    out = ctypes.create_string_buffer(cap)
    mask = ctypes.create_string_buffer(cap)
    n = c_size_t(0)
    rc = lib().oracle_cascaded_compress(data, len(data), type_tag, CASCADED_TYPE_SIZE[type_tag], num_rles, num_deltas,
                                        use_bp, ctypes.cast(out, c_void_p), ctypes.cast(mask, c_void_p), ctypes.byref(n))
    if rc != 0:
        raise ValueError("oracle_cascaded_compress: unsupported options")
    return out.raw[: n.value], mask.raw[: n.value]


def cascaded_decompressed_size(comp: bytes) -> int:
    return lib().oracle_cascaded_decompressed_size(comp, len(comp))


def cascaded_decompress(comp: bytes, capacity: int):
    out = ctypes.create_string_buffer(max(capacity, 1))
    n = c_size_t(0)
    st = lib().oracle_cascaded_decompress(comp, len(comp), ctypes.cast(out, c_void_p), capacity, ctypes.byref(n))
    return st, out.raw[: n.value]


def masked_equal(a: bytes, b: bytes, mask: bytes) -> bool:
    if len(a) != len(b) or len(a) != len(mask):
        return False
    import numpy as np
    x = np.frombuffer(a, dtype=np.uint8)
    y = np.frombuffer(b, dtype=np.uint8)
    m = np.frombuffer(mask, dtype=np.uint8)
    return bool(np.all((x & m) == (y & m)))
