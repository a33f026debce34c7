"""Bench-sized batches against the REFERENCE's own kernels (oracle/_ref) on the same
GPU: 20 000 chunks x 64 KiB of every distribution bench.py measures -- the whole
compressed buffer must be the reference's, size for size and byte for byte (the
SHA-256 of both is in the assert message), and decode back to the input.

LZ4: uniform / harness / runs (bench.gen_data) and TPC-H-like text x CHAR / INT,
a batch that mixes chunks without matches and chunks that compress, and three batches whose
chunks the routing kernel misjudges (bench.gen_misrouted: text with random bytes exactly where the
sampler looks goes through the LDS shape, random bytes with text there through the sparse far class).
Snappy: the config-4 text.  Cascaded: the config-3 sorted columns (the reference's
output has don't-care bytes there -- SURVEY.md App. C.4 -- so: sizes, both
decoders on both streams, and byte equality under the oracle's mask on a sample).

Skipped, with the reason, where oracle/_ref is absent."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CHUNKS = int(os.environ.get("HIPCOMP_BULK_CHUNKS", "20000"))   # (profiles/r04_bulk_parity_100000.log: the same tests at 100 000)
SEEDS = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}


def _need_ref(reflib):
    if reflib is None:
        pytest.skip("oracle/_ref/libhipcomp_ref.so absent: nothing to compare the bulk batches with")


def _data(dist, dev):
    import torch
    import bench
    if dist == "text":
        return torch.from_numpy(bench.gen_text(CHUNKS * bench.CHUNK)).to(dev)
    if dist == "mixed":   # chunk i: uniform for even i, harness for odd i
        a = bench.gen_data("uniform", 0, CHUNKS // 2, dev, SEEDS["uniform"]).view(CHUNKS // 2, bench.CHUNK)
        b = bench.gen_data("harness", 0, CHUNKS // 2, dev, SEEDS["harness"]).view(CHUNKS // 2, bench.CHUNK)
        return torch.stack([a, b], dim=1).reshape(-1).contiguous()
    if dist.startswith("misrouted_"):   # the bytes the routing kernel looks at misrepresent the chunk
        return bench.gen_misrouted(dist[len("misrouted_"):], min(CHUNKS, 8192), dev)
    return bench.gen_data(dist, 0, CHUNKS, dev, SEEDS[dist])


def _packed(job):
    """the compressed bytes of every chunk, everything behind a chunk's size zeroed"""
    import torch
    n, stride = job.n, job.comp.stride
    A = job.comp.data[: n * stride].view(n, stride)
    idx = torch.arange(stride, device=A.device)[None, :] < job.comp.sizes[:, None]
    return torch.where(idx, A, torch.zeros_like(A))


def _sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def _compare_whole_buffers(mine, ref, what):
    import torch
    assert bool(torch.equal(mine.comp.sizes, ref.comp.sizes)), f"{what}: compressed sizes differ from the reference's"
    a, b = _packed(mine), _packed(ref)
    if not bool(torch.equal(a, b)):
        bad = int((a != b).any(dim=1).nonzero()[0].item())
        raise AssertionError(f"{what}: bytes differ from the reference's, first at chunk {bad}; "
                             f"sha256 mine {_sha(a)} reference {_sha(b)}")
    sha = _sha(a)
    assert sha == _sha(b)
    print(f"{what}: {mine.n} chunks, {int(mine.comp.sizes.sum().item())} compressed bytes, sha256 {sha} == reference build")


@pytest.mark.parametrize("dtype", ["CHAR", "INT"])
@pytest.mark.parametrize("dist", ["uniform", "harness", "runs", "text", "mixed", "misrouted_text_random_samples",
                                  "misrouted_random_text_samples", "misrouted_text_random_first"])
def test_lz4_bulk_batches_equal_the_reference_build(hc, reflib, cuda, dist, dtype):
    import torch
    import bench
    _need_ref(reflib)
    data = _data(dist, cuda)
    opts = hc.LZ4Opts(hc.hipcompType.CHAR if dtype == "CHAR" else hc.hipcompType.INT)
    mine = bench.CodecJob(hc, hc.default_library(), "LZ4", opts, data)
    ref = bench.CodecJob(hc, reflib, "LZ4", opts, data)
    mine.comp.data.zero_()
    ref.comp.data.zero_()
    mine.compress()
    ref.compress()
    torch.cuda.synchronize()
    _compare_whole_buffers(mine, ref, f"LZ4 {dist} {dtype}")
    mine.decompress()
    torch.cuda.synchronize()
    mine.verify()


def test_snappy_bulk_text_equals_the_reference_build(hc, reflib, cuda):
    import torch
    import bench
    _need_ref(reflib)
    data = _data("text", cuda)
    mine = bench.CodecJob(hc, hc.default_library(), "Snappy", hc.SnappyOpts(0), data)
    ref = bench.CodecJob(hc, reflib, "Snappy", hc.SnappyOpts(0), data)
    mine.comp.data.zero_()
    ref.comp.data.zero_()
    mine.compress()
    ref.compress()
    torch.cuda.synchronize()
    _compare_whole_buffers(mine, ref, "Snappy TPC-H-like text")
    mine.decompress()
    torch.cuda.synchronize()
    mine.verify()


def test_cascaded_bulk_columns_against_the_reference_build(hc, oracle, reflib, cuda):
    import torch
    import bench
    _need_ref(reflib)
    data = bench.gen_sorted_columns(CHUNKS, cuda)
    opts = hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1)
    mine = bench.CodecJob(hc, hc.default_library(), "Cascaded", opts, data)
    ref = bench.CodecJob(hc, reflib, "Cascaded", opts, data)
    mine.comp.data.zero_()
    ref.comp.data.zero_()
    mine.compress()
    ref.compress()
    torch.cuda.synchronize()
    assert bool(torch.equal(mine.comp.sizes, ref.comp.sizes)), "Cascaded: compressed sizes differ from the reference's"
    # each decoder on the other's streams
    for enc, dec, who in ((mine, ref, "reference decodes ours"), (ref, mine, "ours decodes the reference's")):
        dec.out.data.zero_()
        st = dec.codec.decompress_async(enc.comp, dec.caps, dec.actual, dec.dtemp, dec.out, dec.statuses)
        assert st == 0
        torch.cuda.synchronize()
        dec.verify()
    # byte equality wherever the format defines the byte: a sample against the oracle's mask
    host = data.cpu().numpy()
    a, b = _packed(mine), _packed(ref)
    sizes = mine.comp.sizes.cpu().tolist()
    dont_care = 0
    for i in list(range(0, CHUNKS, CHUNKS // 256))[:256]:
        want, mask = oracle.cascaded_compress(host[i * bench.CHUNK:(i + 1) * bench.CHUNK].tobytes(), 5, 2, 1, 1)
        assert len(want) == sizes[i]
        assert a[i, : sizes[i]].cpu().numpy().tobytes() == want, f"partition {i}: kernel != oracle"
        assert oracle.masked_equal(b[i, : sizes[i]].cpu().numpy().tobytes(), want, mask), f"partition {i}: oracle != reference"
        dont_care += mask.count(b"\x00")
    assert dont_care > 0
    # and everywhere: the two builds differ in no more bytes than the sampled don't-care rate allows
    differ = int((a != b).sum().item())
    total = int(mine.comp.sizes.sum().item())
    assert differ <= total // 20, (differ, total)
