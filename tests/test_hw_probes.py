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
    """(mask, slot-of-lane): LZ4-like patterns -- slots spread over a 16K-entry
    table with forced collisions, plus dense small tables."""
    import random
    rnd = random.Random(11)
    yield (1 << 64) - 1, [t % 37 for t in range(64)]
    yield sum(1 << t for t in (33, 40, 57)), [300] * 64      # a "Lo empty" LZ4 group
    yield sum(1 << t for t in range(32, 61)), [(t % 5) * 100 for t in range(64)]
    yield sum(1 << t for t in range(0, 31)), [((t * 7) % 11) * 50 for t in range(64)]
    for nslots in (16, 64, 1024, 16384, 16384, 16384):
        for _ in range(40):
            mask = rnd.getrandbits(64) | rnd.getrandbits(64)
            slots = [rnd.randrange(nslots) for _ in range(64)]
            for _ in range(10):
                slots[rnd.randrange(64)] = slots[rnd.randrange(64)]
            yield mask, slots


def _run(lib, space, mask, slots, nslots, cuda):
    import torch
    out = torch.full((nslots,), 0xFFFF, dtype=torch.int32, device=cuda).to(torch.int16)
    slot = torch.tensor(slots, dtype=torch.int32, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if space == "lds8":
        rc = lib.probe_lds_store_byte(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                      ctypes.c_ulonglong(mask), nslots, st)
    elif space == "global":
        rc = lib.probe_global_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                          ctypes.c_ulonglong(mask), st)
    else:
        rc = lib.probe_lds_store_short(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                       ctypes.c_ulonglong(mask), nslots, st)
    assert rc == 0
    torch.cuda.synchronize()
    return [v & 0xFFFF for v in out.cpu().tolist()]


@pytest.mark.parametrize("space", ["global", "lds", "lds8"])
def test_same_address_store_winner(cuda, oracle, space):
    """global_store_short: survivor = last lane in the measured write order
    (oracle.gfx950_store_order_key) -- what the reference's hash-table insert
    gets from the hardware.  ds_write_b16: survivor = highest lane -- what this
    library's table insert (lz4_kernels.hip: sigma order) relies on.  ds_write_b8:
    the same lane -- the tag table is written next to the position table by the
    same lanes and has to keep the tag of the lane whose position survives."""
    lib = _lib()
    nslots = 16384
    groups = bad = 0
    for mask, slots in _cases():
        got = _run(lib, space, mask, slots, nslots, cuda)
        by_slot = {}
        for t in range(64):
            if (mask >> t) & 1:
                by_slot.setdefault(slots[t], []).append(t)
        for s, lanes in by_slot.items():
            w = got[s] - 1000
            assert w in lanes
            if len(lanes) > 1:
                groups += 1
                want = max(lanes, key=oracle.gfx950_store_order_key) if space == "global" else max(lanes)
                bad += (w != want)
    print(f"[probe] {space}: {groups} multi-lane groups, {bad} off-rule")
    assert groups > 500 and bad == 0


@pytest.mark.parametrize("nwords", [40960, 12288, 256])
def test_lds_store_beyond_the_allocation_is_dropped(cuda, nwords):
    """The LZ4 encoder's walk sends the table stores of lanes that must not
    store to LDS address 0x30000, beyond the 160 KiB a workgroup can own
    (lz4_kernels.hip: kLdsNowhere), when its tables leave no byte for a scratch
    slot: the hardware drops such stores and reads from there return 0."""
    import torch
    lib = _lib()
    out = torch.zeros(nwords + 64, dtype=torch.int32, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.probe_lds_out_of_range(ctypes.c_void_p(out.data_ptr()), nwords, ctypes.c_uint32(0x30000), st)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype("uint32")
    import numpy as np
    assert (got[:nwords] == 0xA5000000 + np.arange(nwords, dtype=np.uint32)).all()   # nothing inside changed
    assert (got[nwords:] == 0).all()                                                # reads from nowhere: 0
