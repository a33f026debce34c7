"""Quick LZ4 compress/decompress timing of the built library on one GPU:
   quick_lz4.py [--chunks N] [--dist uniform|harness|runs|text] [--dtype char|int] [--check]
--check compares every chunk's compressed bytes with the reference build (oracle/_ref) when present."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=20000)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--dtype", default="char")
ap.add_argument("--check", action="store_true")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lib", default=None, help="another build of the library (measurement variants)")
ap.add_argument("--no-verify", action="store_true")
ap.add_argument("--count-sequences", action="store_true", help="LZ4 sequences per chunk (mean of the first 64 chunks)")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
dev = torch.device("cuda:0")
for dist in a.dist.split(","):
    for dt in a.dtype.split(","):
        if dist == "text":
            data = torch.from_numpy(bench.gen_text(a.chunks * bench.CHUNK)).to(dev)
        else:
            data = bench.gen_data(dist, 0, a.chunks, dev, {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}[dist])
        t = hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT
        lib = hc.HipcompLibrary(os.path.join(ROOT, a.lib)) if a.lib else hc.default_library()
        job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(t), data)
        job.compress(); job.decompress(); torch.cuda.synchronize()
        if not a.no_verify:
            job.verify()
        tc, td = bench.time_phases(job, a.reps)
        nb, cb = job.total, job.compressed_bytes()
        line = f"{dist:8s} {dt:4s} n={job.n}: compress {min(tc):8.3f} ms {nb/min(tc)/1e6:8.1f} GB/s | decompress {min(td):8.3f} ms {nb/min(td)/1e6:8.1f} GB/s | ratio {nb/cb:.3f}"
        if a.check:
            from oracle import oracle as O
            if os.path.exists(O.REF_LIB_PATH):
                rjob = bench.CodecJob(hc, hc.HipcompLibrary(O.REF_LIB_PATH), "LZ4", hc.LZ4Opts(t), data)
                rjob.compress(); torch.cuda.synchronize()
                same = bool(torch.equal(rjob.comp.sizes, job.comp.sizes))
                if same:
                    # compare the bytes of every chunk up to its size
                    stride = job.comp.stride
                    idx = torch.arange(stride, device=dev)[None, :] < job.comp.sizes[:, None]
                    A = job.comp.data[: job.n * stride].view(job.n, stride)
                    B = rjob.comp.data[: job.n * stride].view(job.n, stride)
                    same = bool(((A == B) | ~idx).all().item())
                line += f" | same_as_reference={same}"
                del rjob
        if a.count_sequences:
            def sequences(blk: bytes) -> int:
                i, n = 0, 0
                while i < len(blk):
                    tok = blk[i]; i += 1
                    lit = tok >> 4
                    if lit == 15:
                        while True:
                            b = blk[i]; i += 1; lit += b
                            if b != 255: break
                    i += lit
                    n += 1
                    if i >= len(blk): break
                    i += 2
                    if (tok & 15) == 15:
                        while True:
                            b = blk[i]; i += 1
                            if b != 255: break
                return n
            k = min(64, job.n)
            sizes = job.comp.sizes[:k].cpu().tolist()
            host = job.comp.data[: k * job.comp.stride].cpu().numpy()
            tot = sum(sequences(host[i * job.comp.stride: i * job.comp.stride + sizes[i]].tobytes()) for i in range(k))
            line += f" | sequences_per_chunk={tot / k:.1f}"
        print(line, flush=True)
        del job, data
        torch.cuda.empty_cache()
