#!/bin/bash
# round-5 expansion experiments: A/B of the fill modes on the bench workload, then the split form under the kernel trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp_r5
mkdir -p $OUT
cd $R
CONFIGS="${CONFIGS:-base32:fill_mode=0,fill_chunk=32;res32:fill_mode=1,fill_chunk=32;res16:fill_mode=1,fill_chunk=16;res32w6:fill_mode=1,fill_chunk=32,fill_waves=6;res64:fill_mode=1,fill_chunk=64;split64k16:fill_mode=2,fill_chunk=64,fill_dense_k=16;split32k16:fill_mode=2,fill_chunk=32,fill_dense_k=16;split64k8:fill_mode=2,fill_chunk=64,fill_dense_k=8;split64k32:fill_mode=2,fill_chunk=64,fill_dense_k=32;split64k64:fill_mode=2,fill_chunk=64,fill_dense_k=64;base32b:fill_mode=0,fill_chunk=32}" \
  timeout 900 python3 tools/ab_t6.py > $OUT/ab.txt 2>&1
tail -30 $OUT/ab.txt
if [ -n "$TRACE_CONFIGS" ]; then
  cd /tmp
  CONFIGS="$TRACE_CONFIGS" timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/ab_t6.py > $OUT/trace.txt 2>&1
  python3 $R/tools/prof_summary.py $OUT/trace/t_results.db > $OUT/trace_stats.txt 2>&1 || true
  head -40 $OUT/trace_stats.txt
fi
