"""Per-window (or per-sequence) PMC figures of the LZ4 compress kernel from rocprofv3 --pmc passes.
usage: pmc_per_window.py [--decompress | --kernel NAME] <units per dispatch> <dir> [<dir> ...]   (dirs hold *_counter_collection.csv)
Prints counter value / units, averaged over the dispatches of the compress (or decompress) kernel that did
the work (with the shape picked per call, the launch of the other shape leaves at once: it is left out)."""
import csv, glob, os, re, sys
which = "lz4_compress"
if sys.argv[1] == "--decompress":
    which = "lz4_decompress"
    del sys.argv[1]
elif sys.argv[1] == "--kernel":      # e.g. --kernel snappy_compress
    which = sys.argv[2]
    del sys.argv[1:3]
units = float(sys.argv[1])
acc = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                m = re.search(which + r"_kernel\w*(<[^>]*>)?", r["Kernel_Name"])
                if m:
                    acc.setdefault(r["Counter_Name"], {}).setdefault(m.group(0), []).append(float(r["Counter_Value"]))
# the kernel that did the work: the one with the most wave cycles (else the largest mean of the counter);
# of its dispatches those that did work (a compress call launches one kernel per class and the ones
# whose list is empty leave at once: dispatches below a tenth of the largest are left out)
ref = acc.get("SQ_WAVE_CYCLES") or acc.get("SQ_INSTS_VALU") or next(iter(acc.values()))
kernel = max(ref, key=lambda k: max(ref[k]))
for c in sorted(acc):
    if kernel not in acc[c]:
        continue
    v = acc[c][kernel]
    top = max(v)
    w = [x for x in v if x * 10 >= top] if top > 0 else v
    print(f"{c:28s} {sum(w)/len(w)/units:10.2f}   ({kernel}, dispatches {len(w)} of {len(v)})")
