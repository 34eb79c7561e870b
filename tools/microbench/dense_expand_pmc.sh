#!/bin/bash
# the four formulations of tools/microbench/dense_expand.hip: times, then instruction counters per launch (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dense_expand
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
$R/tools/microbench/dense_expand 200000 > $OUT/dense_expand.txt 2>&1
cat $OUT/dense_expand.txt
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_$i -o p -- $R/tools/microbench/dense_expand 200000 > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 > $OUT/dense_expand_pmc.json
python3 - <<PY
import json
d=json.load(open("$OUT/dense_expand_pmc.json"))
for k,v in d.items():
    print(k[:60], {c: round(x/1e6,2) if x>1e5 else round(x,1) for c,x in v.items()})
PY
