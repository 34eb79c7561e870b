#!/bin/bash
# A/B of engine builds (VS_ENGINE_LIB): the plan's light apply step, the region copy folded into k_t6_bounds, the 32-bit-word
# expansion at 8 waves per SIMD.  LIBS="old apply cur" (files under variantstore_amd/lib/ab/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp_r5d
mkdir -p $OUT
cd $R
export HOSTTIME=1
for rep in 1 2; do
for lib in ${LIBS:-old cur}; do
  export VS_ENGINE_LIB=$R/variantstore_amd/lib/ab/$lib.so
  echo "== $lib chr1 whole batch / shard 3/8" | tee -a $OUT/ab.txt
  CONFIGS="$lib:share_lists=1" timeout 600 python3 tools/ab_t6.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  SHARD=3/8 CONFIGS="$lib-shard:share_lists=1" timeout 600 python3 tools/ab_t6.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
done
for lib in ${LIBS:-old cur}; do
  export VS_ENGINE_LIB=$R/variantstore_amd/lib/ab/$lib.so
  echo "== $lib tcga" | tee -a $OUT/ab.txt
  WL=tcga-10k CONFIGS="$lib-tcga:share_lists=1" timeout 900 python3 tools/ab_t6.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
unset VS_ENGINE_LIB
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r5d -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --latency-samples 0 --extras none > $OUT/trace.log 2>&1
python3 tools/trace_gaps.py $(ls $R/gpurun_out/prof_r5d/*.db | head -1) | tee $OUT/gaps.txt
