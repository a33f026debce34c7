"""The whole-array run-length / delta / bit-packing primitives (hipcomp/primitives.h,
reference src/{RunLengthEncodeGPU,DeltaGPU,BitPackGPU}.h) against host loops, on the data
of the reference's unit tests (src/test/*_test.cpp: runs of random length over small
alphabets, rand()%3==0 -> new value below 1024) for all eight integer types."""
import ctypes

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
NP = datagen.CASCADED_NP


def _lib(hc):
    L = ctypes.CDLL(hc.default_library().path)
    return L


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
    return out


@pytest.mark.parametrize("t", range(8))
def test_run_length_encode(hc, cuda, t):
    import torch
    L = _lib(hc)
    rng = np.random.default_rng(50 + t)
    for ct in (3, 5, 7):                      # ushort, uint, ulonglong counts
        for a in _test_arrays(t, rng):
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
            if n:
                starts = np.flatnonzero(np.concatenate([[True], a[1:] != a[:-1]]))
                want_vals = a[starts]
                want_cnts = np.diff(np.concatenate([starts, [n]]))
            else:
                want_vals, want_cnts = a, np.array([], dtype=np.int64)
            assert runs == want_vals.size
            got_vals = d_vals.cpu().numpy().view(a.dtype)[:runs]
            got_cnts = d_cnts.cpu().numpy().view(NP[ct])[:runs]
            assert (got_vals == want_vals).all()
            assert (got_cnts == want_cnts.astype(NP[ct])).all()
            # the downstream form: element count and output addresses live on the device
            if n:
                d_vals2 = torch.zeros_like(d_vals)
                d_cnts2 = torch.zeros_like(d_cnts)
                ptrs = torch.tensor([d_vals2.data_ptr(), d_cnts2.data_ptr()], dtype=torch.int64, device=cuda)
                d_n = torch.tensor([n - 1], dtype=torch.int64, device=cuda)       # one fewer than the buffer holds
                d_num2 = torch.zeros(1, dtype=torch.int64, device=cuda)
                st = L.hipcompRunLengthEncodeCompressDownstream(
                    ctypes.c_void_p(work.data_ptr()), ctypes.c_size_t(work.numel()), t, ctypes.c_void_p(ptrs.data_ptr()),
                    ct, ctypes.c_void_p(ptrs.data_ptr() + 8), ctypes.c_void_p(d_num2.data_ptr()),
                    ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_n.data_ptr()), ctypes.c_size_t(n), _stream())
                assert st == 0
                torch.cuda.synchronize()
                b = a[: n - 1]
                starts = np.flatnonzero(np.concatenate([[True], b[1:] != b[:-1]])) if n > 1 else np.array([], dtype=np.int64)
                assert int(d_num2.item()) == starts.size
                assert (d_vals2.cpu().numpy().view(a.dtype)[: starts.size] == b[starts]).all()
                assert (d_cnts2.cpu().numpy().view(NP[ct])[: starts.size]
                        == np.diff(np.concatenate([starts, [n - 1]])).astype(NP[ct])).all()


@pytest.mark.parametrize("t", range(8))
def test_delta(hc, cuda, t):
    import torch
    L = _lib(hc)
    rng = np.random.default_rng(60 + t)
    for a in _test_arrays(t, rng):
        n = a.size
        if n == 0:
            continue
        d_in = _dev(a, cuda)
        d_out = torch.zeros(n * a.itemsize, dtype=torch.uint8, device=cuda)
        ptr = torch.tensor([d_out.data_ptr()], dtype=torch.int64, device=cuda)
        d_n = torch.tensor([n], dtype=torch.int64, device=cuda)
        st = L.hipcompDeltaCompress(None, ctypes.c_size_t(0), t, ctypes.c_void_p(ptr.data_ptr()),
                                    ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_n.data_ptr()),
                                    ctypes.c_size_t(n), _stream())
        assert st == 0
        torch.cuda.synchronize()
        with np.errstate(over="ignore"):
            want = a.copy()
            want[1:] = a[1:] - a[:-1]          # wrap-around in the element type
        assert (d_out.cpu().numpy().view(a.dtype) == want).all()


@pytest.mark.parametrize("t", range(8))
def test_bit_pack(hc, cuda, t):
    import torch
    L = _lib(hc)
    rng = np.random.default_rng(70 + t)
    dt = NP[t]
    wbits = 64 if dt().itemsize == 8 else 32
    for a in _test_arrays(t, rng) + [np.full(777, 5, dtype=dt)]:
        n = a.size
        if n == 0:
            continue
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
        lo, hi = int(a.min()), int(a.max())
        assert int(d_min.cpu().numpy().view(dt)[0]) == lo
        bits = (hi - lo).bit_length()
        assert int(d_bits[0].item()) == bits
        if bits == 0:
            continue
        words = (n * bits + wbits - 1) // wbits
        got = d_out.cpu().numpy().view(np.uint64 if wbits == 64 else np.uint32)[:words]
        big = 0
        for i, v in enumerate(a.tolist()):       # value i at bit i * bits, least significant first
            big |= (v - lo) << (i * bits)
        want = [(big >> (w * wbits)) & ((1 << wbits) - 1) for w in range(words)]
        assert got.tolist() == want


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
