#!/bin/bash
# Hardware counters of the bench command, one rocprofv3 --pmc pass per counter group (never combined with tracing):
#   tools/pmc_collect.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# Output under gpurun_out/pmc_<tag>_<i>/; summarise with tools/pmc_summary.py.
# The program after `--` is python3 itself (no env/bash hops: the profiler has initialised the GPU by then).
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for group in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- \
    python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --latency-samples 0 --skip-extras > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i ($group): rc=$?"
done
