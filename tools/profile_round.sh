#!/bin/bash
# One profiling round of the bench command on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [workload]
#  1. rocprofv3 --kernel-trace --stats (headline batch alone; then with the type-4, point-query and sample-coordinate legs)
#                                                 -> <tag>_kernel_stats.txt, <tag>_kernel_stats_t4.txt
#  2. rocprofv3 --pmc, one pass per counter group -> <tag>_pmc.json, traffic_<workload>.json
#  3. the bench line of the same tree             -> <tag>_bench.json
# Everything lands in gpurun_out/profiles_<tag>/ (the only directory that travels back); copy what is to be kept into profiles/.
# The program after `--` is python3 itself (no env/bash hops: the profiler has initialised the GPU by then);
# counters are never combined with tracing.
tag=$1; wl=${2:-chr1-2504}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
ARGS0="--steps 6 --warmup 2 --no-cpu-baseline --latency-samples 0 --extras none --workload $wl"
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --latency-samples 0 --extras t4,points,sc --workload $wl"
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernels_hash())" > $O/kernels_blob.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py $ARGS0 > $R/gpurun_out/prof_$tag.log 2>&1
echo "kernel-trace rc=$?"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_$tag/p_results.db "rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS0" > $O/${tag}_kernel_stats.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_t4 -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${tag}_t4.log 2>&1
echo "kernel-trace (with type 4) rc=$?"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_${tag}_t4/p_results.db "rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS" > $O/${tag}_kernel_stats_t4.txt
i=0
for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pmc pass $i ($group): rc=$?"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4 > $O/${tag}_pmc.json
python3 tools/make_traffic_json.py $wl $tag --blob=$(cat $O/kernels_blob.txt) gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4
cp profiles/traffic_$wl.json $O/
python3 bench.py --workload $wl > $O/${tag}_bench.json 2> gpurun_out/${tag}_bench_err.log
echo "bench rc=$?"
python3 bench.py --workload $wl --extras pipelined --no-cpu-baseline --latency-samples 0 > $O/${tag}_bench_pipelined.json 2>> gpurun_out/${tag}_bench_err.log
echo "bench (pipelined leg) rc=$?"
tail -c 600 $O/${tag}_bench.json
