"""Diagnostic: dump same-address store winners (global_store_short / ds_write_b16)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tests", "probes", "libhwprobes.so"))
dev = torch.device("cuda:0")
import random
random.seed(1)
cases = []
cases.append(((1 << 64) - 1, [0] * 64))
cases.append(((1 << 64) - 1, [t % 7 for t in range(64)]))
cases.append((sum(1 << t for t in (33, 40, 57)), [3] * 64))
cases.append((sum(1 << t for t in range(32, 61)), [t % 5 for t in range(64)]))
cases.append((sum(1 << t for t in range(0, 31)), [(t * 7) % 11 for t in range(64)]))
cases.append((0x8000000100000001, [9] * 64))
for _ in range(6):
    cases.append((random.getrandbits(64), [random.randrange(6) for _ in range(64)]))
for space in ("global", "lds"):
    for ci, (mask, slots) in enumerate(cases):
        out = torch.full((64,), 0xFFFF, dtype=torch.int32, device=dev).to(torch.int16)
        slot = torch.tensor(slots, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if space == "global":
            lib.probe_global_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()), ctypes.c_ulonglong(mask), st)
        else:
            lib.probe_lds_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()), ctypes.c_ulonglong(mask), 64, st)
        torch.cuda.synchronize()
        got = [v & 0xFFFF for v in out.cpu().tolist()]
        for s in range(64):
            lanes = [t for t in range(64) if (mask >> t) & 1 and slots[t] == s]
            if len(lanes) > 1:
                w = got[s] - 1000
                tag = "HIGH" if w == max(lanes) else "LOW" if w == min(lanes) else "OTHER"
                print(space, "case", ci, "slot", s, "lanes", lanes, "winner", w, tag)
