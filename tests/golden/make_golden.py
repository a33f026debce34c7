"""Generate golden vectors from the REFERENCE's own kernels.

Runs on the MI355X box only (needs a GPU and oracle/_ref/libhipcomp_ref.so,
the reference's low-level sources compiled unmodified by oracle/Makefile):

    gpurun -- python tests/golden/make_golden.py gpurun_out/golden

Inputs are regenerated from tests/datagen.py (seeded), so the fixture holds
only the reference's OUTPUT: full compressed bytes for small chunks, SHA-256 +
length for large ones.  The committed results are tests/golden/{lz4,snappy,cascaded}_reference.json.
"""
import base64
import hashlib
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

FULL_BYTES_LIMIT = 2048


def cases():
    import datagen
    for name, data in datagen.edge_chunks():
        yield "edge/" + name, data
    for bi, chunks in enumerate(datagen.harness_batches()):
        if bi >= 3:
            break
        for ci, c in enumerate(chunks):
            yield f"harness/b{bi}/c{ci}", c


def record(blob: bytes):
    r = {"len": len(blob), "sha256": hashlib.sha256(blob).hexdigest()}
    if len(blob) <= FULL_BYTES_LIMIT:
        r["b64"] = base64.b64encode(blob).decode()
    return r


def main(outdir):
    import torch
    hc = importlib.import_module("hipcomp-core_amd")
    from oracle import oracle as O
    ref = hc.HipcompLibrary(O.REF_LIB_PATH)
    named = list(cases())
    chunks = [c for _, c in named]
    src = hc.batch.from_host_chunks(chunks, "cuda:0")
    out = {"generator": "tests/golden/make_golden.py", "library": "oracle/_ref/libhipcomp_ref.so (reference build)",
           "device": torch.cuda.get_device_name(0), "lz4": []}
    for tname, dtype, es in (("CHAR", 0, 1), ("USHORT", 3, 2), ("INT", 4, 4)):
        for max_chunk in (65536, 1000, 0):
            comp = hc.batch.Codec("LZ4", hc.LZ4Opts(dtype), lib=ref).compress(src, max_chunk)
            torch.cuda.synchronize()
            got = comp.to_host_chunks()
            for (name, c), g in zip(named, got):
                rec = record(g)
                rec.update({"case": name, "elem_size": es, "max_chunk": max_chunk, "in_len": len(c),
                            "in_sha256": hashlib.sha256(c).hexdigest()})
                out["lz4"].append(rec)
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "lz4_reference.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(out["lz4"]), "lz4 records")

    # ---- Snappy (one configuration: the codec has no options)
    sn = {"generator": out["generator"], "library": out["library"], "device": out["device"], "snappy": []}
    comp = hc.batch.Codec("Snappy", lib=ref).compress(src)
    torch.cuda.synchronize()
    for (name, c), g in zip(named, comp.to_host_chunks()):
        rec = record(g)
        rec.update({"case": name, "in_len": len(c), "in_sha256": hashlib.sha256(c).hexdigest()})
        sn["snappy"].append(rec)
    with open(os.path.join(outdir, "snappy_reference.json"), "w") as f:
        json.dump(sn, f, separators=(",", ":"))
    print("wrote", len(sn["snappy"]), "snappy records")

    # ---- Cascaded: uint32/int16/uint8/int64 x three option sets; the reference's
    # output carries don't-care bytes (stale LDS), so the FULL bytes are stored
    # for every record and compared under the oracle's mask.
    import zlib
    import datagen
    ca = {"generator": out["generator"], "library": out["library"], "device": out["device"],
          "inputs": "tests/datagen.py:cascaded_golden_inputs(type)", "cascaded": []}
    for t in (5, 2, 1, 6):
        named_c = datagen.cascaded_golden_inputs(t)
        csrc = hc.batch.from_host_chunks([c for _, c in named_c], "cuda:0")
        for (R, D, bp) in ((2, 1, 1), (2, 1, 0), (1, 0, 1)):
            comp = hc.batch.Codec("Cascaded", hc.CascadedOpts(4096, t, R, D, bp), lib=ref).compress(csrc)
            torch.cuda.synchronize()
            for (name, c), g in zip(named_c, comp.to_host_chunks()):
                ca["cascaded"].append({"case": name, "type": t, "opts": [R, D, bp],
                                       "in_sha256": hashlib.sha256(c).hexdigest(), "out_len": len(g),
                                       "out_zb64": base64.b64encode(zlib.compress(g, 9)).decode()})
    with open(os.path.join(outdir, "cascaded_reference.json"), "w") as f:
        json.dump(ca, f, separators=(",", ":"))
    print("wrote", len(ca["cascaded"]), "cascaded records")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
