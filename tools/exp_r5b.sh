#!/bin/bash
# A/B of engine builds (VS_ENGINE_LIB) on the bench workload: CONFIGS per build, then optional kernel trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp_r5b
mkdir -p $OUT
cd $R
for lib in $LIBS; do
  name=$(basename $lib .so)
  echo "== $name" | tee -a $OUT/ab.txt
  if [ "$lib" = "main" ]; then unset VS_ENGINE_LIB; else export VS_ENGINE_LIB=$R/$lib; fi
  timeout 900 python3 tools/ab_t6.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
