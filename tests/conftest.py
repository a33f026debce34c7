import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hc():
    """The product package (its directory name has a hyphen)."""
    return importlib.import_module("hipcomp-core_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def reflib(hc):
    """The reference's own low-level build (oracle/_ref), or None if absent."""
    from oracle import oracle as O
    if not os.path.exists(O.REF_LIB_PATH):
        return None
    return hc.HipcompLibrary(O.REF_LIB_PATH)


LZ4_SHAPES = ("auto", "mix", "pair", "far", "fars", "farw")


def force_lz4_shape(hc, monkeypatch, shape):
    """Makes hc.batch.Codec(...) without a `lib` use the library that `shape` needs: the product
    (lib/libhipcomp.so: it reads nothing from the environment, every chunk goes where the routing
    kernel sends it) for "auto", the knobs build (lib/libhipcomp_knobs.so: the same sources and the
    same device code -- tests/test_build_guards_cpu.py -- with HIPCOMP_LZ4_SHAPE read at every call)
    for a forced shape."""
    monkeypatch.delenv("HIPCOMP_LZ4_PAIR", raising=False)
    if shape == "auto":
        monkeypatch.delenv("HIPCOMP_LZ4_SHAPE", raising=False)
        return
    # "pair": the LDS shape as two waves per chunk (lz4_mix.hiph, lz4_compress_kernel_pair) whatever the size of
    # the batch -- the product takes it from 1536 chunks of 32 .. 64 KiB on; "mix": four lone waves per workgroup
    monkeypatch.setenv("HIPCOMP_LZ4_PAIR", "1" if shape == "pair" else "0")
    monkeypatch.setenv("HIPCOMP_LZ4_SHAPE", "mix" if shape == "pair" else shape)
    monkeypatch.setattr(hc.batch, "default_library", hc.knobs_library)


@pytest.fixture(params=LZ4_SHAPES)
def lz4_shape(request, monkeypatch, hc):
    """Every launch shape of the LZ4 encoder in turn: "auto" (the product library) lets the routing
    kernel send every chunk to the shape its data calls for, the others ("pair": the LDS shape with two
    waves per chunk; the knobs build, see
    force_lz4_shape) force one for all chunks -- also on data it would never be picked for (the far
    shapes on chunks without a match, the LDS shape and the wide form on text).  The compressed
    bytes must not depend on it."""
    force_lz4_shape(hc, monkeypatch, request.param)
    return request.param


def compare_with_reference(reflib, what, check):
    """Runs `check(reflib)` -- the asserts of a test that compare with the
    reference's own build -- or, when oracle/_ref is absent, reports the test
    as SKIPPED with the reason instead of passing silently.  Call it last: the
    asserts against the oracle have passed by then."""
    if reflib is None:
        pytest.skip(f"oracle/_ref/libhipcomp_ref.so absent: {what} not compared with the reference build "
                    "(the comparisons with the oracle passed)")
    check(reflib)


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    torch.cuda.set_device(0)
    return torch.device("cuda:0")
