"""The high-level interface's managers (hipcomp/{lz4,snappy,cascaded}.hpp and the factory, via
the C binding hipcomp/hlif.h):
container layout (reference src/hipcomp_common_deps/hlif_shared_types.hpp:68-84,
src/highlevel/BatchManager.hpp:108-112), chunks = the batched API's streams = the oracle's,
round trips, and both directions against the REFERENCE's own manager (oracle/_ref/hlif_ref_tool,
the reference's unmodified high-level sources driven on files)."""
import ctypes
import os
import struct
import subprocess
from ctypes import c_int, c_size_t, c_void_p

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "hlif_ref_tool")


def _lib(hc):
    return ctypes.CDLL(hc.default_library().path)


class CascadedOpts(ctypes.Structure):
    _fields_ = [("chunk_size", c_size_t), ("type", c_int), ("num_RLEs", c_int), ("num_deltas", c_int), ("use_bp", c_int)]


class Manager:
    def __init__(self, L, chunk=None, dtype=None, snappy=False, cascaded=None, container=None, cuda=None):
        self.L = L
        self.h = c_void_p()
        if container is not None:                         # the factory: manager from a container on the device
            import torch
            self.keep = torch.from_numpy(np.frombuffer(container, dtype=np.uint8).copy()).to(cuda)
            assert L.hipcompHlifManagerCreateFromContainer(c_void_p(self.keep.data_ptr()), None, ctypes.byref(self.h)) == 0
        elif cascaded is not None:
            L.hipcompHlifCascadedManagerCreate.argtypes = [CascadedOpts, c_void_p, c_void_p]
            assert L.hipcompHlifCascadedManagerCreate(cascaded, None, ctypes.byref(self.h)) == 0
        elif snappy:
            assert L.hipcompHlifSnappyManagerCreate(c_size_t(chunk), None, ctypes.byref(self.h)) == 0
        else:
            assert L.hipcompHlifLZ4ManagerCreate(c_size_t(chunk), c_int(dtype), None, ctypes.byref(self.h)) == 0

    def close(self):
        self.L.hipcompHlifManagerDestroy(self.h)

    def compress(self, data: bytes, cuda):
        import torch
        mx, nc = c_size_t(0), c_size_t(0)
        assert self.L.hipcompHlifConfigureCompression(self.h, c_size_t(len(data)), ctypes.byref(mx), ctypes.byref(nc)) == 0
        src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(cuda) if data else torch.empty(8, dtype=torch.uint8, device=cuda)
        dst = torch.zeros(mx.value + 8, dtype=torch.uint8, device=cuda)
        assert self.L.hipcompHlifCompress(self.h, c_void_p(src.data_ptr()), c_size_t(len(data)), c_void_p(dst.data_ptr())) == 0
        st = c_int(-1)
        assert self.L.hipcompHlifGetLastStatus(self.h, ctypes.byref(st)) == 0 and st.value == 0
        size = c_size_t(0)
        assert self.L.hipcompHlifGetCompressedSize(self.h, c_void_p(dst.data_ptr()), ctypes.byref(size)) == 0
        assert size.value <= mx.value
        return dst[: size.value].cpu().numpy().tobytes(), nc.value

    def decompress(self, container: bytes, cuda):
        import torch
        src = torch.from_numpy(np.frombuffer(container, dtype=np.uint8).copy()).to(cuda)
        n, nc = c_size_t(0), c_size_t(0)
        assert self.L.hipcompHlifGetDecompressedSize(self.h, c_void_p(src.data_ptr()), ctypes.byref(n), ctypes.byref(nc)) == 0
        dst = torch.zeros(max(n.value, 8), dtype=torch.uint8, device=cuda)
        assert self.L.hipcompHlifDecompress(self.h, c_void_p(src.data_ptr()), c_void_p(dst.data_ptr())) == 0
        st = c_int(-1)
        assert self.L.hipcompHlifGetLastStatus(self.h, ctypes.byref(st)) == 0
        return st.value, dst[: n.value].cpu().numpy().tobytes()


def _parse(container: bytes):
    magic, major, minor, fmt = struct.unpack_from("<IBBB", container, 0)
    comp_size, decomp_size, n = struct.unpack_from("<QQQ", container, 8)
    chunk, = struct.unpack_from("<Q", container, 48)
    data_off, = struct.unpack_from("<I", container, 56)
    dtype, = struct.unpack_from("<I", container, 64)
    offs = struct.unpack_from(f"<{n}Q", container, 72)
    sizes = struct.unpack_from(f"<{n}Q", container, 72 + 8 * n)
    return dict(magic=magic, version=(major, minor), format=fmt, comp_size=comp_size, decomp_size=decomp_size, n=n,
                chunk=chunk, data_off=data_off, dtype=dtype, offs=offs, sizes=sizes)


def _inputs():
    rng = np.random.default_rng(5)
    return [datagen.text_like(4, 300001), datagen.harness_like_int32(6, 70000).tobytes(),
            bytes(rng.integers(0, 256, 200000, dtype=np.uint8)), b"abc", b"", bytes(65536 * 3)]


@pytest.mark.parametrize("chunk,dtype,es", [(65536, 0, 1), (4096, 0, 1), (65536, 4, 4), (100000, 2, 2)])
def test_container_layout_chunks_and_round_trip(hc, oracle, cuda, chunk, dtype, es):
    L = _lib(hc)
    m = Manager(L, chunk, dtype)
    for data in _inputs():
        if len(data) % es:
            data = data[: len(data) // es * es]
        cont, nc = m.compress(data, cuda)
        h = _parse(cont)
        assert nc == (len(data) + chunk - 1) // chunk == h["n"]
        assert (h["magic"], h["version"], h["format"]) == (0, (2, 2), 0)
        assert (h["decomp_size"], h["chunk"], h["dtype"]) == (len(data), chunk, dtype)
        assert h["data_off"] == 72 + 24 * nc and len(cont) == h["data_off"] + h["comp_size"]
        assert h["comp_size"] == sum(h["sizes"])
        # packed, in the order the chunks were finished (the encoders place them themselves, like the
        # reference's); each chunk = the batched stream
        spans = sorted(zip(h["offs"], h["sizes"]))
        assert not spans or (spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(nc - 1)))
        for i in range(nc):
            piece = data[i * chunk:(i + 1) * chunk]
            at = h["offs"][i]
            blob = cont[h["data_off"] + at: h["data_off"] + at + h["sizes"][i]]
            assert blob == oracle.lz4_compress(piece, es, chunk)
        st, back = m.decompress(cont, cuda)
        assert st == 0 and back == data
        if nc:                                                # a damaged chunk fails the whole buffer
            bad = bytearray(cont)
            bad[h["data_off"]] ^= 0xF0
            st, _ = m.decompress(bytes(bad), cuda)
            assert st in (0, 12)                              # (a changed token can still decode to the right length)
    m.close()


def test_hostile_container_headers_are_refused(hc, cuda):
    """The container header is the buffer's own word about itself: one that contradicts the
    manager (chunk count that is not ceil(size / chunk), another chunk size or format, data that
    does not start behind its own tables, a larger output than the chunks can fill) must end in
    hipcompErrorCannotDecompress with nothing written past the caller's output -- not in slices
    that wrap around (caps = decomp_bytes - at underflowed)."""
    import torch
    L = _lib(hc)
    chunk = 4096
    m = Manager(L, chunk, 0)
    data = datagen.text_like(31, 10 * chunk - 77)
    cont, nc = m.compress(data, cuda)
    assert nc == 10
    st, back = m.decompress(cont, cuda)
    assert st == 0 and back == data

    def patched(fmt, at, value):
        b = bytearray(cont)
        struct.pack_into(fmt, b, at, value)
        return bytes(b)
    hostile = {
        "num_chunks too large": patched("<Q", 24, 1000),
        "num_chunks too small": patched("<Q", 24, 3),
        "decomp size smaller than the chunks": patched("<Q", 16, 2 * chunk),
        "decomp size larger than the chunks": patched("<Q", 16, 50 * chunk),
        "another chunk size": patched("<Q", 48, 2 * chunk),
        "another format": patched("<B", 6, 1),
        "data offset moved": patched("<I", 56, 72 + 24 * 10 + 4096),
    }
    for what, bad in hostile.items():
        src = torch.from_numpy(np.frombuffer(bad, dtype=np.uint8).copy()).to(cuda)
        # an output buffer of the TRUE size with a guard zone behind it
        dst = torch.full((len(data) + (1 << 20),), 0x5A, dtype=torch.uint8, device=cuda)
        n, k = c_size_t(123), c_size_t(456)
        assert L.hipcompHlifGetDecompressedSize(m.h, c_void_p(src.data_ptr()), ctypes.byref(n), ctypes.byref(k)) == 0
        assert (n.value, k.value) == (0, 0), what
        assert L.hipcompHlifDecompress(m.h, c_void_p(src.data_ptr()), c_void_p(dst.data_ptr())) == 0
        st = c_int(-1)
        assert L.hipcompHlifGetLastStatus(m.h, ctypes.byref(st)) == 0
        assert st.value == 12, what
        assert bool((dst == 0x5A).all().item()), what + ": wrote to the output"
    # the manager still works afterwards
    st, back = m.decompress(cont, cuda)
    assert st == 0 and back == data
    m.close()


def test_many_chunks_take_several_slabs(hc, cuda):
    L = _lib(hc)
    m = Manager(L, 1024, 0)                                   # (one pass; several: the *_placement_takes_several_passes tests)
    data = (datagen.text_like(9, (1 << 20) + 13) * 70)[: 70000 * 1024 - 100]
    cont, nc = m.compress(data, cuda)
    assert nc == 70000
    st, back = m.decompress(cont, cuda)
    assert st == 0 and back == data
    need = c_size_t(0)
    assert L.hipcompHlifGetRequiredScratchBytes(m.h, ctypes.byref(need)) == 0 and need.value > 0
    m.close()
    # the same with the caller's scratch buffer of exactly the required size, guard bytes behind it
    import torch
    m = Manager(L, 1024, 0)
    scratch = torch.full((need.value + 4096,), 0x5A, dtype=torch.uint8, device=cuda)
    assert L.hipcompHlifSetScratchBuffer(m.h, c_void_p(scratch.data_ptr())) == 0
    cont2, _ = m.compress(data, cuda)
    a, b = _parse(cont), _parse(cont2)
    # (the LZ4 encoders place their chunks themselves, in the order they finish -- as the reference's do: the
    # same sizes, every chunk the same bytes wherever it went, the places a gapless tiling of the data)
    assert a["sizes"] == b["sizes"] and a["comp_size"] == b["comp_size"] == sum(a["sizes"])
    for c, p in ((cont, a), (cont2, b)):
        spans = sorted(zip(p["offs"], p["sizes"]))
        assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    body = lambda c, p, i: c[p["data_off"] + p["offs"][i]: p["data_off"] + p["offs"][i] + p["sizes"][i]]
    assert all(body(cont, a, i) == body(cont2, b, i) for i in range(0, a["n"], 97))
    st, back = m.decompress(cont2, cuda)
    assert st == 0 and back == data
    assert bool((scratch[need.value:] == 0x5A).all().item())       # nothing written behind the required size
    m.close()


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference build of the high-level interface not present")
@pytest.mark.parametrize("chunk,dtype", [(65536, 0), (8192, 4)])
def test_containers_interchange_with_the_reference_manager(hc, cuda, tmp_path, chunk, dtype):
    L = _lib(hc)
    m = Manager(L, chunk, dtype)
    data = datagen.text_like(21, 1000000) + datagen.harness_like_int32(3, 50000).tobytes()
    # ours -> reference
    cont, _ = m.compress(data, cuda)
    (tmp_path / "ours.bin").write_bytes(cont)
    r = subprocess.run([REF_TOOL, "decompress", str(tmp_path / "ours.bin"), str(tmp_path / "ours.out")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ours.out").read_bytes() == data
    # reference -> ours
    (tmp_path / "in.bin").write_bytes(data)
    r = subprocess.run([REF_TOOL, "compress", "lz4", str(chunk), str(dtype), str(tmp_path / "in.bin"), str(tmp_path / "ref.bin")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    ref_cont = (tmp_path / "ref.bin").read_bytes()
    st, back = m.decompress(ref_cont, cuda)
    assert st == 0 and back == data
    a, b = _parse(cont), _parse(ref_cont)                     # same header and sizes; the reference places chunks in completion order
    assert {k: a[k] for k in a if k != "offs"} == {k: b[k] for k in b if k != "offs"}
    for i in range(a["n"]):                                  # chunk for chunk the same bytes, placed elsewhere
        assert (cont[a["data_off"] + a["offs"][i]: a["data_off"] + a["offs"][i] + a["sizes"][i]]
                == ref_cont[b["data_off"] + b["offs"][i]: b["data_off"] + b["offs"][i] + b["sizes"][i]])
    m.close()


def _head(container: bytes, fh_bytes: int):
    """common header + arrays of a container whose format header has fh_bytes bytes"""
    magic, major, minor, fmt = struct.unpack_from("<IBBB", container, 0)
    comp_size, decomp_size, n = struct.unpack_from("<QQQ", container, 8)
    chunk, = struct.unpack_from("<Q", container, 48)
    data_off, = struct.unpack_from("<I", container, 56)
    at = (64 + fh_bytes + 7) // 8 * 8
    offs = struct.unpack_from(f"<{n}Q", container, at)
    sizes = struct.unpack_from(f"<{n}Q", container, at + 8 * n)
    return dict(version=(major, minor), format=fmt, comp_size=comp_size, decomp_size=decomp_size, n=n, chunk=chunk,
                data_off=data_off, arrays_at=at, offs=offs, sizes=sizes)


def _ref(args, tmp_path):
    r = subprocess.run([REF_TOOL] + [str(a) for a in args], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("chunk", [65536, 5000])
def test_snappy_manager(hc, oracle, cuda, tmp_path, chunk):
    """Snappy container: format 1, one byte of format header, chunks = the batched Snappy
    streams (= the oracle's); round trip; the factory; both directions against the reference."""
    L = _lib(hc)
    m = Manager(L, chunk, snappy=True)
    data = datagen.tpch_lineitem_text(3, 400000) + bytes(np.random.default_rng(1).integers(0, 256, 70001, dtype=np.uint8))
    for d in (data, b"", b"xyz"):
        cont, nc = m.compress(d, cuda)
        h = _head(cont, 1)
        assert (h["version"], h["format"], h["chunk"], h["decomp_size"], h["n"]) == ((2, 2), 1, chunk, len(d), nc)
        assert h["data_off"] == 72 + 24 * nc and len(cont) == h["data_off"] + h["comp_size"]
        for i in range(nc):
            blob = cont[h["data_off"] + h["offs"][i]: h["data_off"] + h["offs"][i] + h["sizes"][i]]
            assert blob == oracle.snappy_compress(d[i * chunk:(i + 1) * chunk])
        st, back = m.decompress(cont, cuda)
        assert st == 0 and back == d
    cont, _ = m.compress(data, cuda)
    f = Manager(L, container=cont, cuda=cuda)                 # factory picks the Snappy manager
    st, back = f.decompress(cont, cuda)
    assert st == 0 and back == data
    f.close()
    if os.path.exists(REF_TOOL):
        (tmp_path / "ours.bin").write_bytes(cont)
        _ref(["decompress", tmp_path / "ours.bin", tmp_path / "ours.out"], tmp_path)
        assert (tmp_path / "ours.out").read_bytes() == data
        (tmp_path / "in.bin").write_bytes(data)
        _ref(["compress", "snappy", chunk, tmp_path / "in.bin", tmp_path / "ref.bin"], tmp_path)
        ref_cont = (tmp_path / "ref.bin").read_bytes()
        st, back = m.decompress(ref_cont, cuda)
        assert st == 0 and back == data
        b = _head(ref_cont, 1)
        assert b["sizes"] == _head(cont, 1)["sizes"]          # chunk for chunk the same streams
    m.close()


@pytest.mark.parametrize("chunk,tname,t,es,r,dl,bp", [(4096, "UINT", 5, 4, 2, 1, 1), (4096, "UCHAR", 1, 1, 1, 0, 1),
                                                      (16384, "USHORT", 3, 2, 1, 1, 1), (8192, "ULONGLONG", 7, 8, 2, 1, 1)])
def test_cascaded_manager(hc, cuda, tmp_path, chunk, tname, t, es, r, dl, bp):
    """Cascaded container: format 4, the options as format header, every chunk one partition of
    the batched codec (4096-byte sub-chunks); round trip; the factory; the reference."""
    L = _lib(hc)
    opts = CascadedOpts(chunk, t, r, dl, bp)
    m = Manager(L, cascaded=opts)
    col = datagen.sorted_column(11, 300000 // es)
    data = (col.astype({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[es])).tobytes()
    data += bytes(np.random.default_rng(2).integers(0, 256, 5000 * es, dtype=np.uint8))
    for d in (data, b"", data[: 3 * es]):
        cont, nc = m.compress(d, cuda)
        h = _head(cont, 24)
        assert (h["version"], h["format"], h["chunk"], h["decomp_size"], h["n"]) == ((2, 2), 4, chunk, len(d), nc)
        assert struct.unpack_from("<Qiiii", cont, 64) == (chunk, t, r, dl, bp)
        assert h["arrays_at"] == 88 and h["data_off"] == 88 + 24 * nc and len(cont) == h["data_off"] + h["comp_size"]
        assert all(o % 8 == 0 for o in h["offs"])
        st, back = m.decompress(cont, cuda)
        assert st == 0 and back == d
    cont, _ = m.compress(data, cuda)
    assert es != 4 or len(cont) < len(data)                   # (the column survives the narrower types only in part)
    f = Manager(L, container=cont, cuda=cuda)
    st, back = f.decompress(cont, cuda)
    assert st == 0 and back == data
    f.close()
    if os.path.exists(REF_TOOL):
        (tmp_path / "ours.bin").write_bytes(cont)
        _ref(["decompress", tmp_path / "ours.bin", tmp_path / "ours.out"], tmp_path)
        assert (tmp_path / "ours.out").read_bytes() == data
        (tmp_path / "in.bin").write_bytes(data)
        _ref(["compress", "cascaded", chunk, t, r, dl, bp, tmp_path / "in.bin", tmp_path / "ref.bin"], tmp_path)
        ref_cont = (tmp_path / "ref.bin").read_bytes()
        st, back = m.decompress(ref_cont, cuda)
        assert st == 0 and back == data
    m.close()


def test_factory_picks_the_lz4_manager_and_refuses_other_formats(hc, cuda):
    L = _lib(hc)
    m = Manager(L, 8192, 4)
    data = datagen.harness_like_int32(8, 30000).tobytes()
    cont, _ = m.compress(data, cuda)
    f = Manager(L, container=cont, cuda=cuda)
    st, back = f.decompress(cont, cuda)
    assert st == 0 and back == data
    f.close()
    m.close()
    import torch
    bad = bytearray(cont)
    bad[6] = 3                                                # GDeflate: a closed format
    t = torch.from_numpy(np.frombuffer(bytes(bad), dtype=np.uint8).copy()).to(cuda)
    h = c_void_p()
    assert L.hipcompHlifManagerCreateFromContainer(c_void_p(t.data_ptr()), None, ctypes.byref(h)) == 10


def test_lz4_placement_takes_several_passes_and_every_kind_of_chunk_end(hc, oracle, cuda):
    """The LZ4 manager's encoders place their chunks themselves (lz4_launch.hpp, Lz4Placement), up to 262 144
    chunks per pass: 600 000 chunks of 512 bytes are three passes.  Chunks that end in a literal run (placed
    before that run is written), in a match (placed after), incompressible ones, empty tails: sizes = the
    batched streams', places tile the data, the round trip."""
    L = _lib(hc)
    rng = np.random.default_rng(77)
    chunk = 512
    piece = (datagen.text_like(21, 1 << 20) + bytes(rng.integers(0, 256, 1 << 20, dtype=np.uint8))
             + bytes(1 << 19) + datagen.harness_like_int32(22, 1 << 17).tobytes())
    data = (piece * 200)[: 600000 * chunk - 77]
    m = Manager(L, chunk, 0)
    cont, nc = m.compress(data, cuda)
    assert nc == 600000
    h = _parse(cont)
    assert h["comp_size"] == sum(h["sizes"]) and len(cont) == h["data_off"] + h["comp_size"]
    order = np.argsort(np.asarray(h["offs"], dtype=np.int64), kind="stable")
    offs, sizes = np.asarray(h["offs"], dtype=np.int64)[order], np.asarray(h["sizes"], dtype=np.int64)[order]
    assert offs[0] == 0 and bool((offs[:-1] + sizes[:-1] == offs[1:]).all())
    for i in list(range(0, nc, 9973)) + [nc - 1, 262143, 262144, 524287, 524288]:
        blob = cont[h["data_off"] + h["offs"][i]: h["data_off"] + h["offs"][i] + h["sizes"][i]]
        assert blob == oracle.lz4_compress(data[i * chunk:(i + 1) * chunk], 1, chunk), i
    st, back = m.decompress(cont, cuda)
    assert st == 0 and back == data
    m.close()


@pytest.mark.parametrize("codec", ["snappy", "cascaded"])
def test_snappy_and_cascaded_placement_takes_several_passes(hc, oracle, cuda, codec):
    """The Snappy and Cascaded managers' encoders place their chunks themselves too (csrc/placement.hpp: a grid
    as large as the device holds workgroups, chunks off a ticket counter, a slot per workgroup): 600 000 chunks
    of 512 bytes are three passes.  Sizes and bytes = the batched streams' (the oracle's), places tile the data
    (Cascaded: at multiples of 8), the round trip -- with the caller's scratch buffer of exactly the required
    size and guard bytes behind it."""
    import torch
    L = _lib(hc)
    rng = np.random.default_rng(78)
    chunk = 512
    if codec == "snappy":
        piece = (datagen.text_like(23, 1 << 20) + bytes(rng.integers(0, 256, 1 << 19, dtype=np.uint8)) + bytes(1 << 18))
        m = Manager(L, chunk, snappy=True)
        fh, align = 1, 1
        want = lambda b: oracle.snappy_compress(b)
    else:
        piece = (datagen.sorted_column(12, 1 << 18).astype(np.uint32).tobytes()
                 + bytes(rng.integers(0, 256, 1 << 19, dtype=np.uint8)) + bytes(1 << 18))
        m = Manager(L, cascaded=CascadedOpts(chunk, 5, 2, 1, 1))
        fh, align = 24, 8
        want = lambda b: oracle.cascaded_compress(b, 5, 2, 1, 1)[0]
    data = (piece * (600000 * chunk // len(piece) + 1))[: 600000 * chunk - 76]
    need = c_size_t(0)
    assert L.hipcompHlifGetRequiredScratchBytes(m.h, ctypes.byref(need)) == 0 and need.value > 0
    scratch = torch.full((need.value + 4096,), 0x5A, dtype=torch.uint8, device=cuda)
    assert L.hipcompHlifSetScratchBuffer(m.h, c_void_p(scratch.data_ptr())) == 0
    cont, nc = m.compress(data, cuda)
    assert nc == 600000
    h = _head(cont, fh)
    assert len(cont) == h["data_off"] + h["comp_size"]
    offs, sizes = np.asarray(h["offs"], dtype=np.int64), np.asarray(h["sizes"], dtype=np.int64)
    order = np.argsort(offs, kind="stable")
    room = (sizes + align - 1) // align * align
    assert offs[order][0] == 0 and bool((offs[order][:-1] + room[order][:-1] == offs[order][1:]).all())
    assert h["comp_size"] == int(room.sum())
    for i in list(range(0, nc, 9973)) + [nc - 1, 262143, 262144, 524287, 524288]:
        blob = cont[h["data_off"] + h["offs"][i]: h["data_off"] + h["offs"][i] + h["sizes"][i]]
        assert blob == want(data[i * chunk:(i + 1) * chunk]), i
    st, back = m.decompress(cont, cuda)
    assert st == 0 and back == data
    assert bool((scratch[need.value:] == 0x5A).all().item())       # nothing written behind the required size
    m.close()
