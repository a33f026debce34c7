"""profiles/rNN_rows.json from the rocprofv3 passes of scripts/collect_profiles.sh.

usage: profile_rows.py <dir with stats_<K>/ fetch_<K>/ write_<K>/ per row> <json out> <row>=<K> [...]
Per row and phase (compress / decompress): the kernel that did the work (the one with the largest
total time whose name holds the phase), its average duration over the dispatches that did work --
with per-chunk routing a compress call launches one kernel per class and the ones whose list is
empty leave at once: dispatches shorter than a tenth of the longest are left out, here and in the
PMC averages -- and the HBM-side bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE
passes, both ways: traffic_x2 = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-byte
requests at 64 bytes for wide streaming reads, MI355X_MICROARCH.md) and traffic_raw = (FETCH_SIZE +
WRITE_SIZE) x 1024.  The doubling is calibrated on streaming reads; the kernels whose reads are gathers of 2
to 16 bytes anywhere in a table or a chunk -- the LZ4 far kernels and the Snappy encoder -- ask for 64-byte
requests, which FETCH_SIZE counts as they are (profiles/r04_cache_counters.txt), so
traffic_bytes_per_launch -- what bench.py reports -- is the raw figure for those (GATHER_KERNELS) and the
doubled one for all others; "fetch_correction" says which.  The file records the sha256 of the device sources
(bench.kernel_source_id): bench.py ignores it for any other build."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


GATHER_KERNELS = ("lz4_compress_kernel_far", "snappy_compress_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def trace_durations(d):
    """kernel name -> list of dispatch durations (ns) from the kernel trace"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def counter(d, name):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return out


def working(values):
    top = max(values)
    return [v for v in values if v * 10 >= top]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    import bench
    table = {"_how": __doc__.split("\n\n", 1)[1].strip(), "kernel_source_sha16": bench.kernel_source_id(), "device_asm_sha16": bench.device_asm_id(), "rows": {}}
    for spec in sys.argv[3:]:
        row, k = spec.split("=")
        dur = trace_durations(os.path.join(src, "stats_" + k))
        fetch, write = counter(os.path.join(src, "fetch_" + k), "FETCH_SIZE"), counter(os.path.join(src, "write_" + k), "WRITE_SIZE")
        ent = {}
        for phase in ("compress", "decompress"):
            names = [n for n in dur if phase in n and (phase != "compress" or "decompress" not in n)]
            if not names:
                continue
            kn = max(names, key=lambda n: sum(dur[n]))
            w = working(dur[kn])
            e = {"kernel": kn, "avg_ms": sum(w) / len(w) / 1e6, "dispatches": len(w), "dispatches_left_out": len(dur[kn]) - len(w)}
            if kn in fetch and kn in write:
                fw, ww = working(fetch[kn]), working(write[kn])
                f_kib, w_kib = sum(fw) / len(fw), sum(ww) / len(ww)
                gather = any(kn.startswith(g) for g in GATHER_KERNELS)
                e.update({"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                          "traffic_raw_bytes_per_launch": (f_kib + w_kib) * 1024,
                          "traffic_x2_bytes_per_launch": (2 * f_kib + w_kib) * 1024,
                          "fetch_correction": "raw" if gather else "x2",
                          "traffic_bytes_per_launch": ((1 if gather else 2) * f_kib + w_kib) * 1024})
            ent[phase] = e
        table["rows"][row] = ent
    json.dump(table, open(dst, "w"), indent=1)
    print(json.dumps(table["rows"], indent=1))


main()
