"""Pair walk diagnostics (make -C hipcomp-core_amd/csrc VARIANT=pairdbg EXTRA=-DHC_PAIR_DEBUG):
pair_dbg.py [--chunks N] [--dtype char|int] [--pair 1|2]  -- compresses uniform data with the pair kernel, prints the
wait counters, compares the bytes with the old mix kernel's (HIPCOMP_LZ4_PAIR=0)."""
import argparse, ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=2000)
ap.add_argument("--dtype", default="char")
ap.add_argument("--dist", default="uniform")
ap.add_argument("--pair", default="1")
a = ap.parse_args()
hc = importlib.import_module("hipcomp-core_amd")
path = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp_pairdbg.so")
lib = hc.HipcompLibrary(path)
dll = ctypes.CDLL(path)
dev = torch.device("cuda:0")
names = ["walks", "token timeouts", "barrier timeouts", "helper idle timeouts", "stops out of range", "walks ended by a match", "halts (match seen at the token)"]
SEEDS = {"uniform": 0x5EED0002, "harness": 0x5EED0003, "runs": 0x5EED0004}
data = bench.gen_data(a.dist, 0, a.chunks, dev, SEEDS[a.dist])
for dt in a.dtype.split(","):
    t = hc.hipcompType.CHAR if dt == "char" else hc.hipcompType.INT
    job = bench.CodecJob(hc, lib, "LZ4", hc.LZ4Opts(t), data)
    os.environ["HIPCOMP_LZ4_PAIR"] = "0"
    job.compress(); torch.cuda.synchronize()
    ref = (job.comp.sizes.clone(), job.comp.data.clone())
    job.decompress(); torch.cuda.synchronize(); job.verify()
    for mode in a.pair.split(","):
        os.environ["HIPCOMP_LZ4_PAIR"] = mode
        buf = (ctypes.c_uint32 * 32)()
        assert dll.hipcompBatchedLZ4DebugPair(buf, 1) == 0
        job.comp.data.zero_(); job.comp.sizes.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); job.compress(); e1.record(); torch.cuda.synchronize()
        assert dll.hipcompBatchedLZ4DebugPair(buf, 1) == 0
        same_sizes = bool(torch.equal(job.comp.sizes, ref[0]))
        same = same_sizes and bool(torch.equal(job.comp.data, ref[1]))
        nbad = int((job.comp.sizes != ref[0]).sum().item())
        print(f"{a.dist} as {dt}, pair mode {mode}, {job.n} chunks: {e0.elapsed_time(e1):.3f} ms (first call), same_bytes={same} (sizes differ in {nbad} chunks)")
        for k, nme in enumerate(names):
            print(f"   {nme:40s} {buf[k]}")
        if buf[11]:
            print(f"   blocks through the tables {buf[11]}; per block and wave: walk {256 * buf[10] / buf[11]:.0f} ticks, of which waiting for the token {256 * buf[8] / buf[11]:.0f} (s_memtime ticks)")
            w2 = 2 * max(buf[0], 1)
            print(f"   per walk and wave: {256 * buf[10] / w2:.0f} ticks, of which until the first words {256 * buf[13] / w2:.0f}, the end (decisions, barriers, undo) {256 * buf[12] / w2:.0f}")
            print(f"   early looks that did not find the token: {buf[14]} of {buf[11]}")
            print(f"   walks by SIMD: wave 0 {list(buf[16:20])}, wave 1 {list(buf[20:24])}")
        sys.stdout.flush()
    del job
