"""Per kernel, every counter of a set of rocprofv3 --pmc passes as the mean per working dispatch
(dispatches below a tenth of the kernel's largest are launches that found their list empty):
   pmc_table.py <dir> [<dir> ...] [--kernels substr,substr]
Prints `kernel counter mean dispatches` lines, grouped by kernel."""
import csv, glob, os, re, sys
dirs = [d for d in sys.argv[1:] if not d.startswith("--")]
want = None
for i, x in enumerate(sys.argv):
    if x == "--kernels":
        want = sys.argv[i + 1].split(",")
        dirs.remove(sys.argv[i + 1])


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if want and not any(w in k for w in want):
                    continue
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("##", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        top = max(v)
        w = [x for x in v if x * 10 >= top] if top > 0 else v
        print(f"{c:40s} {sum(w) / len(w):18.1f}   ({len(w)} of {len(v)} dispatches)")
