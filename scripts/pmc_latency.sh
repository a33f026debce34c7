set -e
O=gpurun_out/lat
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in VmemLatency LdsLatency InstrFetchLatency; do
rocprofv3 --kernel-trace --output-format csv --pmc $M -d $O/$M -- python3 scripts/quick_lz4.py --chunks 20000 --dist uniform --reps 1 > $O/$M.log 2>&1 || true
done
python3 - <<'PY'
import csv,glob
for m in ["VmemLatency","LdsLatency","InstrFetchLatency"]:
    for f in glob.glob(f"gpurun_out/lat/{m}/**/*counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "lz4" in k[0]: print(m,k,sum(v)/len(v),len(v))
PY
