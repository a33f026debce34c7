"""ctypes binding of the batched codec C ABI (include/hipcomp/*.h).

One :class:`HipcompLibrary` wraps one shared object that exports the 18
``hipcompBatched*`` symbols -- the product ``lib/libhipcomp.so`` by default.
The same class can bind any other library with that ABI (the tests bind the
reference build ``oracle/_ref/libhipcomp_ref.so`` this way to compare bytes on
the GPU); the product never does.

Method names, argument order and meaning are the C functions' (reference
include/hipcomp/lz4.h:106-243, snappy.h:80-195, cascaded.h:142-295).  Device
pointers are passed as plain integers (``tensor.data_ptr()``), streams as the
raw ``hipStream_t`` handle (``torch.cuda.current_stream().cuda_stream``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_int, c_size_t, c_void_p

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in
# the process before libhipcomp.so is loaded so that the library binds to that
# same runtime; otherwise the loader pulls a second copy from /opt/rocm and the
# two runtimes do not share devices or allocations (hipPointerGetAttributes on
# a torch tensor then fails with "no ROCm-capable device").  A C/C++ caller has
# exactly one runtime and needs none of this.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libhipcomp.so")
# the same sources built with -DHC_MEASUREMENT_KNOBS (csrc/Makefile VARIANT=knobs): honours
# HIPCOMP_LZ4_SHAPE / HIPCOMP_LZ4_GEOMETRY / HIPCOMP_LZ4_SPAN -- tests and measurement scripts only
KNOBS_LIB = os.path.join(_HERE, "lib", "libhipcomp_knobs.so")


class hipcompStatus:
    """include/hipcomp/shared_types.h"""

    Success = 0
    ErrorInvalidValue = 10
    ErrorNotSupported = 11
    ErrorCannotDecompress = 12
    ErrorCudaError = 1000
    ErrorInternal = 10000


class hipcompType:
    """include/hipcomp.h"""

    CHAR = 0
    UCHAR = 1
    SHORT = 2
    USHORT = 3
    INT = 4
    UINT = 5
    LONGLONG = 6
    ULONGLONG = 7
    BITS = 0xFF

    _SIZES = {0: 1, 1: 1, 2: 2, 3: 2, 4: 4, 5: 4, 6: 8, 7: 8, 0xFF: 1}

    @classmethod
    def size_of(cls, t: int) -> int:
        return cls._SIZES[t]


class LZ4Opts(ctypes.Structure):
    _fields_ = [("data_type", c_int)]


class SnappyOpts(ctypes.Structure):
    _fields_ = [("reserved", c_int)]


class CascadedOpts(ctypes.Structure):
    _fields_ = [
        ("chunk_size", c_size_t),
        ("type", c_int),
        ("num_RLEs", c_int),
        ("num_deltas", c_int),
        ("use_bp", c_int),
    ]


LZ4_DEFAULT_OPTS = LZ4Opts(hipcompType.CHAR)
SNAPPY_DEFAULT_OPTS = SnappyOpts(0)
CASCADED_DEFAULT_OPTS = CascadedOpts(4096, hipcompType.INT, 2, 1, 1)

_OPTS = {"LZ4": LZ4Opts, "Snappy": SnappyOpts, "Cascaded": CascadedOpts}


def _sigs(codec: str):
    opts = _OPTS[codec]
    p = c_void_p
    return {
        f"hipcompBatched{codec}CompressGetTempSize": [c_size_t, c_size_t, opts, POINTER(c_size_t)],
        f"hipcompBatched{codec}CompressGetMaxOutputChunkSize": [c_size_t, opts, POINTER(c_size_t)],
        f"hipcompBatched{codec}CompressAsync": [p, p, c_size_t, c_size_t, p, c_size_t, p, p, opts, p],
        f"hipcompBatched{codec}DecompressGetTempSize": [c_size_t, c_size_t, POINTER(c_size_t)],
        f"hipcompBatched{codec}DecompressAsync": [p, p, p, p, c_size_t, p, c_size_t, p, p, p],
        f"hipcompBatched{codec}GetDecompressSizeAsync": [p, p, p, c_size_t, p],
    }


ABI_SYMBOLS = tuple(name for codec in _OPTS for name in _sigs(codec))


class HipcompLibrary:
    """A loaded shared object exporting the batched codec C ABI."""

    def __init__(self, path: str = DEFAULT_LIB, codecs=("LZ4", "Snappy", "Cascaded")):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C hipcomp-core_amd/csrc`). There is no fallback path."
            )
        self.path = path
        # RTLD_LOCAL: several libraries with the same symbol names can coexist
        self._dll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        self.codecs = tuple(codecs)
        for codec in self.codecs:
            for name, argtypes in _sigs(codec).items():
                fn = getattr(self._dll, name)  # AttributeError if not exported
                fn.argtypes = argtypes
                fn.restype = c_int
                setattr(self, name, fn)

    # -- size queries as plain Python -----------------------------------
    def compress_temp_size(self, codec: str, batch: int, max_chunk: int, opts) -> int:
        out = c_size_t(0)
        st = getattr(self, f"hipcompBatched{codec}CompressGetTempSize")(batch, max_chunk, opts, ctypes.byref(out))
        if st != 0:
            raise RuntimeError(f"hipcompBatched{codec}CompressGetTempSize -> status {st}")
        return out.value

    def max_output_chunk_size(self, codec: str, max_chunk: int, opts) -> int:
        out = c_size_t(0)
        st = getattr(self, f"hipcompBatched{codec}CompressGetMaxOutputChunkSize")(max_chunk, opts, ctypes.byref(out))
        if st != 0:
            raise RuntimeError(f"hipcompBatched{codec}CompressGetMaxOutputChunkSize -> status {st}")
        return out.value

    def cascaded_select_opts(self, ptrs: int, sizes: int, batch: int, type_tag: int, temp: int, temp_bytes: int, stream: int = 0):
        """hipcompBatchedCascadedSelectOpts (include/hipcomp/cascaded_select.h, an API of this library's own):
        -> (CascadedOpts, estimated ratio).  Synchronises the stream."""
        fn = self._dll.hipcompBatchedCascadedSelectOpts
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, POINTER(CascadedOpts), POINTER(ctypes.c_double), c_void_p]
        opts, ratio = CascadedOpts(), ctypes.c_double(0.0)
        st = fn(ptrs, sizes, batch, type_tag, temp, temp_bytes, ctypes.byref(opts), ctypes.byref(ratio), stream)
        if st != 0:
            raise RuntimeError(f"hipcompBatchedCascadedSelectOpts -> status {st}")
        return opts, ratio.value

    def cascaded_select_temp_size(self) -> int:
        out = c_size_t(0)
        fn = self._dll.hipcompBatchedCascadedSelectOptsGetTempSize
        fn.restype = c_int
        fn.argtypes = [POINTER(c_size_t)]
        assert fn(ctypes.byref(out)) == 0
        return out.value

    def decompress_temp_size(self, codec: str, num_chunks: int, max_chunk: int) -> int:
        out = c_size_t(0)
        st = getattr(self, f"hipcompBatched{codec}DecompressGetTempSize")(num_chunks, max_chunk, ctypes.byref(out))
        if st != 0:
            raise RuntimeError(f"hipcompBatched{codec}DecompressGetTempSize -> status {st}")
        return out.value


_default = None


_knobs = None


def knobs_library() -> HipcompLibrary:
    """The test / measurement build that reads the launch-shape knobs from the environment."""
    global _knobs
    if _knobs is None:
        _knobs = HipcompLibrary(KNOBS_LIB, codecs=_available_codecs(KNOBS_LIB))
    return _knobs


def default_library() -> HipcompLibrary:
    """The product library, loaded once.  Raises ImportError if not built."""
    global _default
    if _default is None:
        _default = HipcompLibrary(DEFAULT_LIB, codecs=_available_codecs(DEFAULT_LIB))
    return _default


def _available_codecs(path: str):
    # During bring-up the library may export a subset of the codecs; bind what
    # is there and let a missing one fail at the call site (AttributeError).
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: run __graft_entry__.build() first. There is no fallback path."
        )
    dll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    out = []
    for codec in _OPTS:
        try:
            getattr(dll, f"hipcompBatched{codec}CompressAsync")
            out.append(codec)
        except AttributeError:
            pass
    return tuple(out)


# Loading at import time makes a missing build fail loudly and early.
default_library()
