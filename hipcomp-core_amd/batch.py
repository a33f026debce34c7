"""Chunk lists in HBM and thin call helpers over the C ABI.

Plumbing only (device memory and streams come from torch); every codec
operation below is one call into ``libhipcomp.so``.  Layout: a batch of chunks
is ONE contiguous ``uint8`` device buffer with a fixed stride per chunk plus
two device arrays the C ABI consumes directly -- chunk addresses (``void*[]``)
and chunk sizes (``size_t[]``), both built on the device so no host transfer
sits inside a timed region.  The stride is a multiple of 16 bytes so every
chunk starts 16-byte aligned (Cascaded requires element alignment; LZ4 typed
modes require ``sizeof(T)`` alignment).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import api
from .api import HipcompLibrary, default_library


def _stream_handle(stream=None) -> int:
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ChunkBatch:
    """``n`` chunks at ``data[i*stride : i*stride + sizes[i]]`` on one device."""

    data: torch.Tensor   # uint8 [n * stride] (device)
    ptrs: torch.Tensor   # int64 [n] device addresses (void*[])
    sizes: torch.Tensor  # int64 [n] bytes (size_t[])
    stride: int

    @property
    def n(self) -> int:
        return int(self.ptrs.numel())

    @property
    def device(self):
        return self.data.device

    def chunk_bytes(self, i: int, size: Optional[int] = None) -> bytes:
        """Host copy of chunk ``i`` (test helper)."""
        if size is None:
            size = int(self.sizes[i].item())
        return self.data[i * self.stride : i * self.stride + size].cpu().numpy().tobytes()

    def to_host_chunks(self):
        sizes = self.sizes.cpu().numpy()
        host = self.data.cpu().numpy()
        return [host[i * self.stride : i * self.stride + int(sizes[i])].tobytes() for i in range(self.n)]


def make_ptrs(data: torch.Tensor, n: int, stride: int) -> torch.Tensor:
    base = data.data_ptr()
    return base + torch.arange(n, device=data.device, dtype=torch.int64) * stride


def alloc_batch(n: int, stride: int, device="cuda", fill: Optional[int] = None) -> ChunkBatch:
    stride = _round_up(max(stride, 1), 16)
    if fill is None:
        data = torch.empty(max(n * stride, 16), dtype=torch.uint8, device=device)
    else:
        data = torch.full((max(n * stride, 16),), fill, dtype=torch.uint8, device=device)
    return ChunkBatch(data, make_ptrs(data, n, stride), torch.zeros(n, dtype=torch.int64, device=device), stride)


def from_host_chunks(chunks: Sequence[bytes], device="cuda", stride: Optional[int] = None) -> ChunkBatch:
    n = len(chunks)
    mx = max([len(c) for c in chunks], default=0)
    stride = _round_up(max(stride or mx, 1), 16)
    host = np.zeros(max(n * stride, 16), dtype=np.uint8)
    for i, c in enumerate(chunks):
        host[i * stride : i * stride + len(c)] = np.frombuffer(c, dtype=np.uint8)
    data = torch.from_numpy(host).to(device)
    sizes = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=device)
    return ChunkBatch(data, make_ptrs(data, n, stride), sizes, stride)


def from_device_buffer(data: torch.Tensor, chunk_bytes: int) -> ChunkBatch:
    """View a contiguous uint8 device buffer as equal chunks (last may be short)."""
    assert data.dtype == torch.uint8 and data.is_contiguous()
    assert chunk_bytes % 16 == 0
    total = data.numel()
    n = (total + chunk_bytes - 1) // chunk_bytes
    sizes = torch.full((n,), chunk_bytes, dtype=torch.int64, device=data.device)
    if n and total % chunk_bytes:
        sizes[-1] = total % chunk_bytes
    return ChunkBatch(data, make_ptrs(data, n, chunk_bytes), sizes, chunk_bytes)


def _check(status: int, what: str):
    if status != api.hipcompStatus.Success:
        raise RuntimeError(f"{what} returned hipcompStatus_t {status}")


class Codec:
    """Calls of one codec ("LZ4", "Snappy" or "Cascaded") on one library."""

    def __init__(self, name: str, opts=None, lib: Optional[HipcompLibrary] = None):
        self.name = name
        self.lib = lib or default_library()
        if opts is None:
            opts = {"LZ4": api.LZ4_DEFAULT_OPTS, "Snappy": api.SNAPPY_DEFAULT_OPTS,
                    "Cascaded": api.CASCADED_DEFAULT_OPTS}[name]
        self.opts = opts
        self._f = lambda suffix: getattr(self.lib, f"hipcompBatched{name}{suffix}")

    # -- size queries ----------------------------------------------------
    def compress_temp_size(self, batch: int, max_chunk: int) -> int:
        return self.lib.compress_temp_size(self.name, batch, max_chunk, self.opts)

    def max_output_chunk_size(self, max_chunk: int) -> int:
        return self.lib.max_output_chunk_size(self.name, max_chunk, self.opts)

    def decompress_temp_size(self, num_chunks: int, max_chunk: int) -> int:
        return self.lib.decompress_temp_size(self.name, num_chunks, max_chunk)

    # -- async calls (raw: caller owns every buffer) -----------------------
    def compress_async(self, src: ChunkBatch, max_chunk: int, temp: Optional[torch.Tensor],
                       dst: ChunkBatch, stream=None) -> int:
        return self._f("CompressAsync")(
            _ptr(src.ptrs), _ptr(src.sizes), max_chunk, src.n,
            _ptr(temp), 0 if temp is None else temp.numel(),
            _ptr(dst.ptrs), _ptr(dst.sizes), self.opts, _stream_handle(stream))

    def decompress_async(self, comp: ChunkBatch, out_caps: torch.Tensor, actual: Optional[torch.Tensor],
                         temp: Optional[torch.Tensor], dst: ChunkBatch,
                         statuses: Optional[torch.Tensor], stream=None) -> int:
        return self._f("DecompressAsync")(
            _ptr(comp.ptrs), _ptr(comp.sizes), _ptr(out_caps), _ptr(actual), comp.n,
            _ptr(temp), 0 if temp is None else temp.numel(),
            _ptr(dst.ptrs), _ptr(statuses), _stream_handle(stream))

    def get_decompress_size_async(self, comp: ChunkBatch, sizes_out: torch.Tensor, stream=None) -> int:
        return self._f("GetDecompressSizeAsync")(
            _ptr(comp.ptrs), _ptr(comp.sizes), _ptr(sizes_out), comp.n, _stream_handle(stream))

    # -- convenience (allocates like a caller of the C API would) ----------
    def compress(self, src: ChunkBatch, max_chunk: Optional[int] = None) -> ChunkBatch:
        """``max_chunk`` is the value handed to the C API as
        max_uncompressed_chunk_bytes (for LZ4 it sizes the hash table and so
        takes part in the result); output buffers are always sized from the
        real largest chunk."""
        real_max = int(src.sizes.max().item()) if src.n else 0
        if max_chunk is None:
            max_chunk = real_max
        dst = alloc_batch(src.n, self.max_output_chunk_size(max(real_max, max_chunk)), src.device)
        tbytes = self.compress_temp_size(src.n, max_chunk)
        temp = torch.empty(max(tbytes, 8), dtype=torch.uint8, device=src.device)
        _check(self.compress_async(src, max_chunk, temp, dst), f"hipcompBatched{self.name}CompressAsync")
        return dst

    def decompress(self, comp: ChunkBatch, out_capacity: int, with_status: bool = True):
        dev = comp.device
        dst = alloc_batch(comp.n, out_capacity, dev)
        caps = torch.full((comp.n,), out_capacity, dtype=torch.int64, device=dev)
        actual = torch.full((comp.n,), -1, dtype=torch.int64, device=dev) if with_status else None
        statuses = torch.full((comp.n,), -1, dtype=torch.int32, device=dev) if with_status else None
        tbytes = self.decompress_temp_size(comp.n, out_capacity)
        temp = torch.empty(max(tbytes, 8), dtype=torch.uint8, device=dev)
        _check(self.decompress_async(comp, caps, actual, temp, dst, statuses),
               f"hipcompBatched{self.name}DecompressAsync")
        if actual is not None:
            dst.sizes = actual
        return dst, actual, statuses

    def get_decompress_size(self, comp: ChunkBatch) -> torch.Tensor:
        out = torch.full((comp.n,), -1, dtype=torch.int64, device=comp.device)
        _check(self.get_decompress_size_async(comp, out), f"hipcompBatched{self.name}GetDecompressSizeAsync")
        return out
