#!/bin/bash
# counters of the split form (lists + rows | dense sites) with the engine build given in VS_ENGINE_LIB
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export CONFIGS='split64k16:fill_mode=2,fill_chunk=64,fill_dense_k=16;base32:fill_mode=0,fill_chunk=32'
bash tools/pmc_any.sh ${1:-split} "tools/ab_t6.py" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM"
python3 - <<'PY'
import json,sys,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
d=json.load(open(f"{R}/gpurun_out/pmc_%s.json" % (sys.argv[1] if len(sys.argv)>1 else "split")))
for k,v in d.items():
    if "k_fill" in k: print(k, json.dumps(v))
PY
