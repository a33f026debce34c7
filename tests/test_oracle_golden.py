"""Pins the CPU oracle to the reference: tests/golden/lz4_reference.json holds
the compressed output of the reference's own kernels (oracle/_ref build, run on
MI355X by tests/golden/make_golden.py) for seeded inputs; the oracle must
reproduce every record byte for byte.  Runs without a GPU."""
import base64
import hashlib
import json
import os

import pytest

import datagen

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "lz4_reference.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def inputs():
    out = {"edge/" + n: d for n, d in datagen.edge_chunks()}
    for bi, chunks in enumerate(datagen.harness_batches()):
        if bi >= 3:
            break
        for ci, c in enumerate(chunks):
            out[f"harness/b{bi}/c{ci}"] = c
    return out


def test_lz4_oracle_reproduces_reference_output(oracle, golden, inputs):
    recs = golden["lz4"]
    assert len(recs) >= 300
    full = 0
    for r in recs:
        data = inputs[r["case"]]
        assert len(data) == r["in_len"] and hashlib.sha256(data).hexdigest() == r["in_sha256"], r["case"]
        got = oracle.lz4_compress(data, r["elem_size"], r["max_chunk"], valid_offsets=False)
        assert len(got) == r["len"], (r["case"], r["elem_size"], r["max_chunk"])
        assert hashlib.sha256(got).hexdigest() == r["sha256"], (r["case"], r["elem_size"], r["max_chunk"])
        if "b64" in r:
            assert got == base64.b64decode(r["b64"])
            full += 1
        # the product's form is identical for chunks <= 64 KiB
        assert got == oracle.lz4_compress(data, r["elem_size"], r["max_chunk"], valid_offsets=True)
    assert full >= 100


def test_snappy_oracle_reproduces_reference_output(oracle, inputs):
    with open(os.path.join(HERE, "golden", "snappy_reference.json")) as f:
        recs = json.load(f)["snappy"]
    assert len(recs) >= 40
    for r in recs:
        data = inputs[r["case"]]
        assert hashlib.sha256(data).hexdigest() == r["in_sha256"], r["case"]
        got = oracle.snappy_compress(data)
        assert len(got) == r["len"], r["case"]
        assert hashlib.sha256(got).hexdigest() == r["sha256"], r["case"]
        if "b64" in r:
            assert got == base64.b64decode(r["b64"])


def test_cascaded_oracle_reproduces_reference_output_under_mask(oracle):
    """The reference's Cascaded output carries don't-care bytes (stale LDS,
    unwritten gaps); every byte the format defines must match the oracle."""
    import zlib
    with open(os.path.join(HERE, "golden", "cascaded_reference.json")) as f:
        recs = json.load(f)["cascaded"]
    assert len(recs) >= 80
    cache = {}
    checked_dont_care = 0
    for r in recs:
        t = r["type"]
        if t not in cache:
            cache[t] = dict(datagen.cascaded_golden_inputs(t))
        data = cache[t][r["case"]]
        assert hashlib.sha256(data).hexdigest() == r["in_sha256"], (r["case"], t)
        ref_out = zlib.decompress(base64.b64decode(r["out_zb64"]))
        assert len(ref_out) == r["out_len"]
        R, D, bp = r["opts"]
        want, mask = oracle.cascaded_compress(data, t, R, D, bp)
        assert len(want) == len(ref_out), (r["case"], t, r["opts"])
        assert oracle.masked_equal(ref_out, want, mask), (r["case"], t, r["opts"])
        checked_dont_care += mask.count(b"\x00")
        # and the oracle decodes the reference's bytes as they are
        s = oracle.CASCADED_TYPE_SIZE[t]
        assert oracle.cascaded_decompress(ref_out, len(data)) == (0, data[: len(data) // s * s])
    assert checked_dont_care > 0
