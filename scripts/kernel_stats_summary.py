"""Short table of a rocprofv3 --kernel-trace --stats run: kernel_stats_summary.py <dir>"""
import csv, glob, os, re, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
        if "hcamd" in name:
            print(f"{name:60s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
