"""Extract the decoder golden vectors the reference's own test holds
(/root/reference/tests/test_snappy_app.cpp:210-223: two compressed streams and
their expected output) into a data fixture.  Runs in the build container only
(the reference tree does not exist on the GPU box).  Output:
tests/golden/snappy_app_vectors.json (hex strings -- data, no source text)."""
import json
import os
import re

SRC = "/root/reference/tests/test_snappy_app.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))


def array(text, name):
    m = re.search(r"uint8_t\s+" + name + r"\[\]\s*=\s*\{([^}]*)\}", text)
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", m.group(1)))


def main():
    text = open(SRC).read()
    out = []
    for i, (csize, usize) in enumerate(((497, 709), (434, 7581)), start=1):
        comp = array(text, f"comp_data{i}")[:csize]
        exp = array(text, f"decomp_data_expected{i}")[:usize]
        assert len(comp) == csize and len(exp) == usize
        out.append({"name": f"test_snappy_app_{i}", "compressed_hex": comp.hex(), "expected_hex": exp.hex()})
    with open(os.path.join(HERE, "snappy_app_vectors.json"), "w") as f:
        json.dump({"source": "reference tests/test_snappy_app.cpp:210-223 (data arrays only)", "vectors": out}, f)
    print("ok", [(len(v["compressed_hex"]) // 2, len(v["expected_hex"]) // 2) for v in out])


if __name__ == "__main__":
    main()
