"""profiles/lz4_hbm_traffic.json from two rocprofv3 PMC passes over bench.py.

usage: hbm_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> <out.json> [label]
The passes are
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 bench.py --no-cpu --no-variants --steps 2 --warmup 1
and the same with WRITE_SIZE (separate runs, as MI355X_MICROARCH.md prescribes).  Counters are KiB
per dispatch, averaged over the dispatches; traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950:
FETCH_SIZE tallies 128-byte requests at 64 bytes for wide streaming reads; the compress kernel's
dword gathers are narrower than that calibrated case, so its read side is an upper estimate).
"""
import csv, glob, json, os, sys


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = "compress" if "lz4_compress_kernel" in r["Kernel_Name"] else "decompress" if "lz4_decompress_kernel" in r["Kernel_Name"] else None
            if k:
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
label = sys.argv[4] if len(sys.argv) > 4 else "uniform/char/100000"
out = {"_how": __doc__.split("\n\n", 1)[1].strip(), label: {
    "kernel": "lz4_compress_kernel<1>",
    "FETCH_SIZE_KiB": fetch["compress"], "WRITE_SIZE_KiB": write["compress"],
    "traffic_bytes_per_launch": (2 * fetch["compress"] + write["compress"]) * 1024,
    "decompress_kernel": {"FETCH_SIZE_KiB": fetch["decompress"], "WRITE_SIZE_KiB": write["decompress"],
                          "traffic_bytes_per_launch": (2 * fetch["decompress"] + write["decompress"]) * 1024}}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out[label], indent=1))
