#!/bin/bash
# Which stage of the Cascaded encoder's 4-byte fast path owns its instructions (gpurun, from the repo root;
# needs lib/libhipcomp_stop{1..5}.so: for k in 1 2 3 4 5: make -C hipcomp-core_amd/csrc VARIANT=stop$k
# EXTRA=-DHC_CASC_STOP_AFTER=$k).  A stage build leaves every sub-chunk behind stage k (1 load + RLE, 2 its
# lengths packed, 3 delta, 4 the second RLE, 5 its lengths packed; the product = 6: values packed + metadata);
# two builds differ by what the stage between them executes.   scripts/pmc_cascaded_stages.sh [parts=20000] [out]
set -e
N=${1:-20000}; O=${2:-gpurun_out/pmc_casc_stages}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PA="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"
for k in 1 2 3 4 5 6; do
  L=hipcomp-core_amd/lib/libhipcomp_stop$k.so; [ $k = 6 ] && L=hipcomp-core_amd/lib/libhipcomp.so
  timeout -k 10 120 python3 scripts/casc_stage_run.py $L $N 5 > $O/time$k.log 2>&1 || { tail -3 $O/time$k.log; exit 1; }
  i=0
  for P in "$PA" "$PB"; do
    i=$((i+1)); rm -rf $O/s${k}p$i
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $O/s${k}p$i -- python3 scripts/casc_stage_run.py $L $N 1 > $O/s${k}p$i.log 2>&1 || { tail -5 $O/s${k}p$i.log; echo "stage $k pass $i failed"; exit 1; }
  done
  python3 scripts/pmc_table.py $O/s${k}p1 $O/s${k}p2 --kernels cascaded_compress > $O/raw$k.txt
  rm -rf $O/s${k}p[0-9]*
done
python3 - <<PY
N=$N; sub=N*16
names={1:"load + RLE 1",2:"pack lengths 1",3:"delta",4:"RLE 2 (values from LDS)",5:"pack lengths 2",6:"pack values + metadata"}
cum={}
for k in range(1,7):
    r={}
    for l in open("$O/raw%d.txt"%k):
        if l.startswith('##'): continue
        p=l.split(); r[p[0]]=float(p[1])/sub
    t=open("$O/time%d.log"%k).read().strip().splitlines()[-1]
    r['ms']=float(t.split('compress')[1].split('ms')[0])
    cum[k]=r
cols=["SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_BRANCH","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_LDS_IDX_ACTIVE","SQ_LDS_BANK_CONFLICT"]
print("# Cascaded encoder, 4-byte fast path, config-3 columns (%d partitions x 16 sub-chunks): per 4 KiB sub-chunk, what each stage adds"%N)
print("# (stage builds: -DHC_CASC_STOP_AFTER=k; row k = build k minus build k-1; ms = the launch with the stages up to k)")
print("%-28s"%"stage"+"".join("%12s"%c.replace("SQ_INSTS_","").replace("SQ_LDS_IDX_ACTIVE","LDS_cycles").replace("SQ_LDS_BANK_CONFLICT","LDS_confl") for c in cols)+"%10s"%"ms upto")
prev={c:0.0 for c in cols}
for k in range(1,7):
    r=cum[k]
    print("%-28s"%names[k]+"".join("%12.1f"%(r.get(c,0)-prev[c]) for c in cols)+"%10.3f"%r['ms'])
    prev={c:r.get(c,0) for c in cols}
print("%-28s"%"all"+"".join("%12.1f"%cum[6].get(c,0) for c in cols)+"%10.3f"%cum[6]['ms'])
PY
