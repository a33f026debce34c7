"""The whole-array run-length / delta / bit-packing primitives (hipcomp/primitives.h,
reference src/{RunLengthEncodeGPU,DeltaGPU,BitPackGPU}.h) against (1) the REFERENCE's own
classes, compiled from its unmodified sources into oracle/_ref/libhipcomp_prims_ref.so
(oracle/Makefile; C wrappers oracle/prims_ref_shim.cpp) and run on the same GPU with the same
inputs -- every output the reference defines must be identical -- and (2) plain host loops.
Data: the reference's unit tests' own (src/test/{RunLengthEncodeGPU,DeltaGPU,BitPackGPU}_test.cpp:
std::srand(0); every third step on average a new value rand() % 1024; a constant array; rand()
modulo the type's maximum), plus empty / single / all-distinct / long-run / full-range arrays,
for all eight integer types.  Where oracle/_ref is absent the tests are reported as skipped,
with the reason, after the host-loop asserts have passed."""
import ctypes
import os

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
NP = datagen.CASCADED_NP


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIMS_REF = os.path.join(ROOT, "oracle", "_ref", "libhipcomp_prims_ref.so")


def _lib(hc):
    L = ctypes.CDLL(hc.default_library().path)
    return L


class _Ref:
    """the reference build's entry points under the product's names"""

    def __init__(self, dll):
        self._dll = dll

    def __getattr__(self, name):
        return getattr(self._dll, "ref_" + name)


@pytest.fixture(scope="module")
def ref_prims(hc):
    if not os.path.exists(PRIMS_REF):
        return None
    return _Ref(ctypes.CDLL(PRIMS_REF, mode=ctypes.RTLD_LOCAL))


def _finish(ref_prims, what):
    if ref_prims is None:
        pytest.skip(f"{PRIMS_REF} absent: {what} not compared with the reference build (the host-loop asserts passed)")


def _reference_unit_test_data(dt, n=10000):
    """reference src/test/RunLengthEncodeGPU_test.cpp:126-134 / DeltaGPU_test.cpp:135-143, same rand() stream"""
    rnd = datagen.glibc_rand(0)
    out = np.empty(n, dtype=np.int64)
    last = 0
    for i in range(n):
        if rnd() % 3 == 0:
            last = rnd() % 1024
        out[i] = last
    return out.astype(dt)


def _dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).to(cuda)


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _test_arrays(t, rng):
    dt = NP[t]
    out = [np.array([], dtype=dt), np.array([7], dtype=dt), np.arange(1, 2000).astype(dt)]
    # reference DeltaGPU_test.cpp:137-145: every third step a new value below 1024
    n = 100000
    v = np.zeros(n, dtype=np.int64)
    change = rng.integers(0, 3, n) == 0
    vals = rng.integers(0, 1024, n)
    cur = 0
    idx = np.flatnonzero(change)
    v[:] = np.repeat(np.concatenate([[0], vals[idx]]), np.diff(np.concatenate([[0], idx, [n]])))[:n]
    out.append(v.astype(dt))
    out.append(np.repeat(rng.integers(0, 100, 3000), rng.integers(1, 70, 3000)).astype(dt))  # long runs over tiles
    out.append(rng.integers(-(1 << 40), 1 << 40, 5000).astype(dt))                             # full range of the type
    out.append(_reference_unit_test_data(dt))                                                  # the reference tests' own stream
    out.append(np.full(10000, 37, dtype=dt))                        # RunLengthEncodeGPU_test.cpp:371-396 (one run)
    rnd = datagen.glibc_rand(0)                                     # BitPackGPU_test.cpp:318-321
    out.append((np.array([rnd() for _ in range(4000)], dtype=np.int64) % min(int(np.iinfo(dt).max), 1 << 62)).astype(dt))
    return out


def _rle(L, cuda, a, t, ct):
    """-> (runs, values[:runs], counts[:runs]) of one library"""
    import torch
    n = a.size
    ws = ctypes.c_size_t(0)
    assert L.hipcompRunLengthEncodeGetWorkspaceSize(ctypes.c_size_t(n), t, ct, ctypes.byref(ws)) == 0
    work = torch.empty(max(ws.value, 8), dtype=torch.uint8, device=cuda)
    d_in = _dev(a, cuda) if n else torch.empty(8, dtype=torch.uint8, device=cuda)
    d_vals = torch.zeros(max(n, 1) * a.itemsize, dtype=torch.uint8, device=cuda)
    d_cnts = torch.zeros(max(n, 1) * NP[ct]().itemsize, dtype=torch.uint8, device=cuda)
    d_num = torch.full((1,), -1, dtype=torch.int64, device=cuda)
    st = L.hipcompRunLengthEncodeCompress(
        ctypes.c_void_p(work.data_ptr()), ctypes.c_size_t(work.numel()), t, ctypes.c_void_p(d_vals.data_ptr()),
        ct, ctypes.c_void_p(d_cnts.data_ptr()), ctypes.c_void_p(d_num.data_ptr()),
        ctypes.c_void_p(d_in.data_ptr()), ctypes.c_size_t(n), _stream())
    assert st == 0
    torch.cuda.synchronize()
    runs = int(d_num.item())
    return runs, d_vals.cpu().numpy().view(a.dtype)[:runs].copy(), d_cnts.cpu().numpy().view(NP[ct])[:runs].copy()


def _rle_downstream(L, cuda, a, t, ct, n_used):
    import torch
    n = a.size
    ws = ctypes.c_size_t(0)
    assert L.hipcompRunLengthEncodeGetWorkspaceSize(ctypes.c_size_t(n), t, ct, ctypes.byref(ws)) == 0
    work = torch.empty(max(ws.value, 8), dtype=torch.uint8, device=cuda)
    d_in = _dev(a, cuda)
    d_vals2 = torch.zeros(n * a.itemsize, dtype=torch.uint8, device=cuda)
    d_cnts2 = torch.zeros(n * NP[ct]().itemsize, dtype=torch.uint8, device=cuda)
    ptrs = torch.tensor([d_vals2.data_ptr(), d_cnts2.data_ptr()], dtype=torch.int64, device=cuda)
    d_n = torch.tensor([n_used], dtype=torch.int64, device=cuda)
    d_num2 = torch.zeros(1, dtype=torch.int64, device=cuda)
    st = L.hipcompRunLengthEncodeCompressDownstream(
        ctypes.c_void_p(work.data_ptr()), ctypes.c_size_t(work.numel()), t, ctypes.c_void_p(ptrs.data_ptr()),
        ct, ctypes.c_void_p(ptrs.data_ptr() + 8), ctypes.c_void_p(d_num2.data_ptr()),
        ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_n.data_ptr()), ctypes.c_size_t(n), _stream())
    assert st == 0
    torch.cuda.synchronize()
    runs = int(d_num2.item())
    return runs, d_vals2.cpu().numpy().view(a.dtype)[:runs].copy(), d_cnts2.cpu().numpy().view(NP[ct])[:runs].copy()


@pytest.mark.parametrize("t", range(8))
def test_run_length_encode(hc, cuda, ref_prims, t):
    L = _lib(hc)
    rng = np.random.default_rng(50 + t)
    compared = 0
    for ct in (3, 5, 7):                      # ushort, uint, ulonglong counts
        for a in _test_arrays(t, rng):
            n = a.size
            runs, got_vals, got_cnts = _rle(L, cuda, a, t, ct)
            if n:
                starts = np.flatnonzero(np.concatenate([[True], a[1:] != a[:-1]]))
                want_vals = a[starts]
                want_cnts = np.diff(np.concatenate([starts, [n]]))
            else:
                want_vals, want_cnts = a, np.array([], dtype=np.int64)
            assert runs == want_vals.size
            assert (got_vals == want_vals).all()
            assert (got_cnts == want_cnts.astype(NP[ct])).all()
            # a run longer than the count type holds wraps in both builds alike (the host loop above
            # casts the same way); the reference needs n > 0
            if ref_prims is not None and n:
                r_runs, r_vals, r_cnts = _rle(ref_prims, cuda, a, t, ct)
                assert r_runs == runs and (r_vals == got_vals).all() and (r_cnts == got_cnts).all(), (t, ct, n)
                compared += 1
            # the downstream form: element count and output addresses live on the device
            if n:
                d_runs, d_vals, d_cnts = _rle_downstream(L, cuda, a, t, ct, n - 1)   # one fewer than the buffer holds
                b = a[: n - 1]
                starts = np.flatnonzero(np.concatenate([[True], b[1:] != b[:-1]])) if n > 1 else np.array([], dtype=np.int64)
                assert d_runs == starts.size
                assert (d_vals == b[starts]).all()
                assert (d_cnts == np.diff(np.concatenate([starts, [n - 1]])).astype(NP[ct])).all()
                if ref_prims is not None and n > 1:
                    r = _rle_downstream(ref_prims, cuda, a, t, ct, n - 1)
                    assert r[0] == d_runs and (r[1] == d_vals).all() and (r[2] == d_cnts).all(), (t, ct, n)
    _finish(ref_prims, "RunLengthEncodeGPU")
    assert compared > 0


def _delta(L, cuda, a, t):
    import torch
    n = a.size
    d_in = _dev(a, cuda)
    d_out = torch.zeros(n * a.itemsize, dtype=torch.uint8, device=cuda)
    ptr = torch.tensor([d_out.data_ptr()], dtype=torch.int64, device=cuda)
    d_n = torch.tensor([n], dtype=torch.int64, device=cuda)
    st = L.hipcompDeltaCompress(None, ctypes.c_size_t(0), t, ctypes.c_void_p(ptr.data_ptr()),
                                ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_n.data_ptr()),
                                ctypes.c_size_t(n), _stream())
    assert st == 0
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(a.dtype).copy()


@pytest.mark.parametrize("t", range(8))
def test_delta(hc, cuda, ref_prims, t):
    L = _lib(hc)
    rng = np.random.default_rng(60 + t)
    for a in _test_arrays(t, rng):
        if a.size == 0:
            continue
        got = _delta(L, cuda, a, t)
        with np.errstate(over="ignore"):
            want = a.copy()
            want[1:] = a[1:] - a[:-1]          # wrap-around in the element type
        assert (got == want).all()
        if ref_prims is not None:
            assert (_delta(ref_prims, cuda, a, t) == got).all(), (t, a.size)
    _finish(ref_prims, "DeltaGPU")


def _bitpack(L, cuda, a, t):
    """-> (min value bytes, bit count, packed bytes incl. slack)"""
    import torch
    n = a.size
    ws = ctypes.c_size_t(0)
    assert L.hipcompBitPackGetWorkspaceSize(ctypes.c_size_t(n), t, ctypes.byref(ws)) == 0
    work = torch.empty(max(ws.value, 8), dtype=torch.uint8, device=cuda)
    d_in = _dev(a, cuda)
    d_out = torch.zeros(n * 8 + 16, dtype=torch.uint8, device=cuda)
    d_min = torch.zeros(8, dtype=torch.uint8, device=cuda)
    d_bits = torch.full((8,), 0xEE, dtype=torch.uint8, device=cuda)
    ptrs = torch.tensor([d_out.data_ptr(), d_min.data_ptr(), d_bits.data_ptr()], dtype=torch.int64, device=cuda)
    d_n = torch.tensor([n], dtype=torch.int64, device=cuda)
    st = L.hipcompBitPackCompress(
        ctypes.c_void_p(work.data_ptr()), ctypes.c_size_t(work.numel()), t, ctypes.c_void_p(ptrs.data_ptr()),
        ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_n.data_ptr()), ctypes.c_size_t(n),
        ctypes.c_void_p(ptrs.data_ptr() + 8), ctypes.c_void_p(ptrs.data_ptr() + 16), _stream())
    assert st == 0
    torch.cuda.synchronize()
    return d_min.cpu().numpy().copy(), int(d_bits[0].item()), d_out.cpu().numpy().copy()


@pytest.mark.parametrize("t", range(8))
def test_bit_pack(hc, cuda, ref_prims, t):
    L = _lib(hc)
    rng = np.random.default_rng(70 + t)
    dt = NP[t]
    wbits = 64 if dt().itemsize == 8 else 32
    for a in _test_arrays(t, rng) + [np.full(777, 5, dtype=dt)]:
        n = a.size
        if n == 0:
            continue
        got_min, got_bits, got_out = _bitpack(L, cuda, a, t)
        lo, hi = int(a.min()), int(a.max())
        assert int(got_min.view(dt)[0]) == lo
        bits = (hi - lo).bit_length()
        assert got_bits == bits
        words = (n * bits + wbits - 1) // wbits
        if bits:
            got = got_out.view(np.uint64 if wbits == 64 else np.uint32)[:words]
            big = 0
            for i, v in enumerate(a.tolist()):       # value i at bit i * bits, least significant first
                big |= (v - lo) << (i * bits)
            want = [(big >> (w * wbits)) & ((1 << wbits) - 1) for w in range(words)]
            assert got.tolist() == want
        if ref_prims is not None:
            r_min, r_bits, r_out = _bitpack(ref_prims, cuda, a, t)
            assert r_bits == got_bits and (r_min[: dt().itemsize] == got_min[: dt().itemsize]).all(), (t, n)
            nbytes = words * (wbits // 8)
            assert (r_out[:nbytes] == got_out[:nbytes]).all(), (t, n, bits)
    _finish(ref_prims, "BitPackGPU")


def test_primitive_errors(hc, cuda):
    L = _lib(hc)
    ws = ctypes.c_size_t(0)
    assert L.hipcompRunLengthEncodeGetWorkspaceSize(ctypes.c_size_t(10), 99, 5, ctypes.byref(ws)) == 10
    assert L.hipcompBitPackGetWorkspaceSize(ctypes.c_size_t(10), 5, None) == 10
    import torch
    buf = torch.zeros(64, dtype=torch.uint8, device=cuda)
    p = ctypes.c_void_p(buf.data_ptr())
    # workspace too small (reference: std::runtime_error -> here a status)
    assert L.hipcompRunLengthEncodeCompress(p, ctypes.c_size_t(1), 5, p, 5, p, p, p, ctypes.c_size_t(1000), _stream()) == 10
