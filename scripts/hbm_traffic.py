"""profiles/lz4_hbm_traffic.json from two rocprofv3 PMC passes over bench.py.

usage: hbm_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> <json to update> <label>
The passes are
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 bench.py --no-cpu --no-variants --steps 2 --warmup 1 [--dist D]
and the same with WRITE_SIZE (separate runs, as MI355X_MICROARCH.md prescribes); label = dist/dtype/chunks
as bench.py looks it up.  Counters are KiB per dispatch, averaged over the dispatches of the compress
kernel that did the work (the launch of the shape the sampling kernel did not pick leaves at once);
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
for wide streaming reads; narrower gathers are not calibrated, so the read side is an upper estimate).
The entry records the sha256 of lz4_kernels.hip it was measured on: bench.py ignores it for any other build.
"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("lz4_compress_kernel" in r["Kernel_Name"] or "lz4_decompress_kernel" in r["Kernel_Name"]):
                acc.setdefault(re.search(r"lz4_\w+(<\w+>)?", r["Kernel_Name"]).group(0), []).append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    comp = max((k for k in mean if "lz4_compress_kernel" in k), key=lambda k: mean[k])
    dec = max((k for k in mean if "lz4_decompress_kernel" in k), key=lambda k: mean[k])
    return comp, mean[comp], mean[dec]


import bench
ck, fetch_c, fetch_d = per_kernel(sys.argv[1], "FETCH_SIZE")
_, write_c, write_d = per_kernel(sys.argv[2], "WRITE_SIZE")
path, label = sys.argv[3], sys.argv[4]
table = json.load(open(path)) if os.path.exists(path) else {}
table["_how"] = __doc__.split("\n\n", 1)[1].strip()
table[label] = {
    "kernel": ck, "kernel_source_sha16": bench.kernel_source_id(),
    "FETCH_SIZE_KiB": fetch_c, "WRITE_SIZE_KiB": write_c,
    "traffic_bytes_per_launch": (2 * fetch_c + write_c) * 1024,
    "decompress_kernel": {"FETCH_SIZE_KiB": fetch_d, "WRITE_SIZE_KiB": write_d,
                          "traffic_bytes_per_launch": (2 * fetch_d + write_d) * 1024}}
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(table[label], indent=1))
