"""LZ4 throughput against chunk size (compress: = hash table size = waves per CU)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
hc = importlib.import_module("hipcomp-core_amd")
lib = hc.default_library()
dev = torch.device("cuda:0")
data = bench.gen_data(sys.argv[1] if len(sys.argv) > 1 else "uniform", 0, 8000, dev, 0x5EED0002)
codec = hc.batch.Codec("LZ4", hc.LZ4Opts(hc.hipcompType.CHAR), lib=lib)
for chunk in (65536, 32768, 16384, 8192, 4096, 2048, 1024):
    src = hc.batch.from_device_buffer(data, chunk)
    comp = hc.batch.alloc_batch(src.n, codec.max_output_chunk_size(chunk), dev)
    temp = torch.empty(max(codec.compress_temp_size(src.n, chunk), 8), dtype=torch.uint8, device=dev)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st = codec.compress_async(src, chunk, temp, comp); e1.record()
        torch.cuda.synchronize(); assert st == 0
        ts.append(e0.elapsed_time(e1))
    t = min(ts[1:])
    out = hc.batch.alloc_batch(src.n, chunk, dev)
    caps = torch.full((src.n,), chunk, dtype=torch.int64, device=dev)
    actual = torch.zeros(src.n, dtype=torch.int64, device=dev)
    stat = torch.zeros(src.n, dtype=torch.int32, device=dev)
    dtemp = torch.empty(max(codec.decompress_temp_size(src.n, chunk), 8), dtype=torch.uint8, device=dev)
    td = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st = codec.decompress_async(comp, caps, actual, dtemp, out, stat); e1.record()
        torch.cuda.synchronize(); assert st == 0
        td.append(e0.elapsed_time(e1))
    assert int(stat.abs().sum().item()) == 0
    d = min(td[1:])
    print(f"chunk {chunk:6d} n {src.n:7d}: compress {t:8.3f} ms {data.numel() / t / 1e6:8.1f} GB/s | "
          f"decompress {d:8.3f} ms {data.numel() / d / 1e6:8.1f} GB/s", flush=True)
