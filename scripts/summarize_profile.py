"""Condense a rocprofv3 --kernel-trace --stats run (csv) into a small table for profiles/."""
import csv, glob, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", ""))[:90]
            rows.append((name, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]),
                         int(r["MinNs"]), int(r["MaxNs"])))
rows.sort(key=lambda x: -x[2])
with open(dst, "w") as out:
    out.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows:
        out.write("%s,%d,%d,%.1f,%.3f,%d,%d\n" % r)
print(open(dst).read())
