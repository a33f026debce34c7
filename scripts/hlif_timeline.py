"""The kernels of ONE LZ4Manager::compress over bench.py's headline buffer, on a time line:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/hlif_timeline.py run
   python3 scripts/hlif_timeline.py show DIR"""
import csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch
    import bench
    hc = importlib.import_module("hipcomp-core_amd")
    d = bench.gen_data("uniform", 0, 100000, torch.device("cuda:0"), 0x5EED0002)
    print(bench.measure_hlif(hc, d, reps=1))
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last compress: from the last header_kernel on
    idx = [i for i, r in enumerate(rows) if "header_kernel" in r["Kernel_Name"]]
    rows = rows[idx[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("hcamd::(anonymous namespace)::", "").replace("void ", "")
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{a:10.1f} {b:10.1f} {b - a:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {name[:70]}")
        if "slab_streams" in name or "decompress" in name:
            break
