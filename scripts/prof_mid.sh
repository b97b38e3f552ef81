#!/usr/bin/env bash
# stall counters of accumulate_mid (p = 16, both load paths) next to accumulate_narrow (p = 8): one pass per counter set
set -u
export OUT=$PWD/gpurun_out/prof_mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_GATE_EN2" "TCC_REQ TCC_HIT TCC_MISS TCC_EA_RDREQ" "TA_BUSY TA_TA_BUSY TD_TD_BUSY TCP_TCP_TA_DATA_STALL_CYCLES" "TCC_TAG_STALL TCC_EA_RDREQ_32B TCC_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for cfg in "narrow8 1 1000000 1000 8" "mid16old 0 100000 1000 16" "mid16lds 1 100000 1000 16"; do
    set -- $cfg
    export ANOFOX_MID_LDS=$2
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p${i}_$1 -o pmc -- $NB $3 $4 $5 ols 2 > $OUT/p${i}_$1.json 2> $OUT/p${i}_$1.err
    echo "pass $i $1 rc=$?"
  done
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('OUT','gpurun_out/prof_mid')
rows=collections.defaultdict(dict)
for f in sorted(glob.glob(out+'/p*_*/**/*counter_collection.csv',recursive=True)):
    cfg=f.split('/prof_mid/')[1].split('/')[0].split('_',1)[1]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'accumulate' not in k: continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in acc.items(): rows[c][cfg]=sum(v)/len(v)
with open(out+'/summary.txt','w') as fo:
    for c in sorted(rows):
        line=c.ljust(34)+'  '.join(f"{k}={rows[c][k]:.4g}" for k in sorted(rows[c]))
        print(line); fo.write(line+'\n')
PY
