"""The batched calls inside a HIP graph: compress + decompress of a batch captured from a stream once and
replayed -- nothing in the calls synchronises, allocates or reads the host, and every kernel of a call is
launched on the caller's stream.  Every replay must give the sizes and the round trip of the plain calls."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["lz4-uniform", "lz4-harness", "snappy-text", "cascaded-sorted"])
def test_compress_and_decompress_replay_from_a_graph(hc, cuda, which):
    import torch
    import bench
    dev = torch.device("cuda:0")
    if which == "lz4-uniform":      # the LDS-table kernel: 4 096 chunks is more than twice what the chip holds -> companion outside a capture
        codec, opts, data = "LZ4", hc.LZ4Opts(0), bench.gen_data("uniform", 0, 4096, dev, 5)
    elif which == "lz4-harness":
        codec, opts, data = "LZ4", hc.LZ4Opts(0), bench.gen_data("harness", 0, 4096, dev, 6)
    elif which == "snappy-text":
        codec, opts, data = "Snappy", hc.SnappyOpts(0), torch.from_numpy(bench.gen_text(1024 * bench.CHUNK)).to(dev)
    else:
        codec, opts, data = "Cascaded", hc.CascadedOpts(4096, 5, 2, 1, 1), bench.gen_sorted_columns(2048, dev)
    job = bench.CodecJob(hc, hc.default_library(), codec, opts, data)
    job.compress(); job.decompress(); torch.cuda.synchronize()
    job.verify()
    want = job.comp.sizes.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            job.compress()
            job.decompress()
    for _ in range(3):
        job.comp.sizes.zero_()
        job.out.data.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(job.comp.sizes, want)
        job.verify()


def test_decompress_then_compress_in_one_graph(hc, cuda):
    """The other order: a compress call captured BEHIND a decompress call's kernels (its ticket counters are
    zeroed by a kernel of its own on the stream, like the decoder's)."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    a = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(0), bench.gen_data("harness", 0, 3000, dev, 8))
    b = bench.CodecJob(hc, hc.default_library(), "LZ4", hc.LZ4Opts(0), bench.gen_data("uniform", 0, 3000, dev, 9))
    a.compress(); a.decompress(); b.compress(); b.decompress(); torch.cuda.synchronize()
    a.verify(); b.verify()
    want = b.comp.sizes.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a.decompress()
            b.compress()
            b.decompress()
    for _ in range(3):
        a.out.data.zero_(); b.comp.sizes.zero_(); b.out.data.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(b.comp.sizes, want)
        a.verify(); b.verify()
