#!/bin/bash
# sweep of the list threshold (VS_LIST_MAX: classes of at most this many carriers are expanded from decoded id lists, denser ones from bit rows)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp_listmax
mkdir -p $OUT
cd $R
for lm in ${LMS:-640 800 1000 1280 1600 2048 2600}; do
  echo "== VS_LIST_MAX=$lm" | tee -a $OUT/ab.txt
  VS_LIST_MAX=$lm CONFIGS="${CONFIGS:-c32:fill_mode=0,fill_chunk=32;c16:fill_mode=0,fill_chunk=16}" timeout 600 python3 tools/ab_t6.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
