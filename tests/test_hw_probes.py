"""Hardware-defined behaviours the bit-exact LZ4 path depends on (SURVEY.md
section 7 step 4): which lane survives when several lanes of one wave store to
one address in a single instruction.  The oracle's STORE_WINNER_HIGHEST and
the kernel's insert rule both assume "highest lane"; this test measures it."""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    return ctypes.CDLL(os.path.join(HERE, "probes", "libhwprobes.so"))


def _cases():
    # (mask, slot-of-lane)
    yield (1 << 64) - 1, [0] * 64                      # all lanes, one address
    yield (1 << 64) - 1, [t % 7 for t in range(64)]    # 7 groups
    yield sum(1 << t for t in (33, 40, 57)), [3] * 64  # a "Lo empty" LZ4 group
    yield sum(1 << t for t in range(32, 61)), [t % 5 for t in range(64)]
    yield sum(1 << t for t in range(0, 31)), [(t * 7) % 11 for t in range(64)]
    yield 0x8000000100000001, [9] * 64                 # lanes 0, 32, 63


@pytest.mark.parametrize("space", ["global", "lds"])
def test_same_address_store_winner_is_highest_lane(cuda, oracle, space):
    import torch
    lib = _lib()
    nslots = 64
    winners = set()
    for mask, slots in _cases():
        out = torch.full((nslots,), 0xFFFF, dtype=torch.int32, device=cuda).to(torch.int16)
        slot = torch.tensor(slots, dtype=torch.int32, device=cuda)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if space == "global":
            rc = lib.probe_global_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                              ctypes.c_ulonglong(mask), st)
        else:
            rc = lib.probe_lds_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                           ctypes.c_ulonglong(mask), nslots, st)
        assert rc == 0
        torch.cuda.synchronize()
        got = [v & 0xFFFF for v in out.cpu().tolist()]
        for s in range(nslots):
            lanes = [t for t in range(64) if (mask >> t) & 1 and slots[t] == s]
            if not lanes:
                assert got[s] == 0xFFFF
                continue
            w = got[s] - 1000
            assert w in lanes
            if len(lanes) > 1:
                winners.add("highest" if w == max(lanes) else "lowest" if w == min(lanes) else "other")
    print(f"[probe] {space} same-address store winner: {winners}")
    assert winners == {"highest"}, winners
    assert oracle.STORE_WINNER_HIGHEST == 1
