#!/bin/bash
# SQ counters of the Cascaded kernels per 4 KiB sub-chunk (gpurun, from the repo root):
#   scripts/pmc_cascaded.sh [parts=20000] [out=gpurun_out/pmc_casc]
set -e
N=${1:-20000}; O=${2:-gpurun_out/pmc_casc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P8="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
P9="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P10="GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P8" "$P9" "$P10"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $O/p$i -- python3 scripts/run_rows.py cascaded/sorted/$N --reps 1 > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; echo "pass $i failed"; }
done
python3 scripts/pmc_table.py $O/p1 $O/p2 $O/p3 --kernels cascaded_compress,cascaded_decompress > $O/raw.txt
rm -rf $O/p[0-9]*
python3 - <<PY
rows={}; k=None
for l in open("$O/raw.txt"):
    if l.startswith('##'): k=l[3:].strip(); rows[k]={}
    else:
        p=l.split(); rows[k][p[0]]=float(p[1])
sub=$N*16
for k,r in rows.items():
    if r.get('SQ_WAVE_CYCLES',0) < 1e6: continue
    print('##',k,'-- per 4 KiB sub-chunk (SQ_WAVE/WAIT/ACTIVE: quad-cycles; LDS_IDX_ACTIVE / BANK_CONFLICT: cycles)')
    for c in sorted(r): print(f'  {c:28s}{r[c]/sub:12.1f}')
    if 'GRBM_GUI_ACTIVE' in r: print(f'  kernel cycles per sub-chunk per CU (GRBM/8 * 256 CUs / sub-chunks): {r["GRBM_GUI_ACTIVE"]/8*256/sub:10.1f}')
PY
