"""Per-window (or per-sequence) PMC figures of the LZ4 compress kernel from rocprofv3 --pmc passes.
usage: pmc_per_window.py <units per dispatch> <dir> [<dir> ...]   (dirs hold *_counter_collection.csv)
Prints counter value / units, averaged over the dispatches of lz4_compress_kernel."""
import csv, glob, os, sys
units = float(sys.argv[1])
acc = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "lz4_compress_kernel" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v)/len(v)/units:10.2f}   (dispatches {len(v)})")
