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
