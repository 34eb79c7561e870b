#!/bin/bash
# Hardware counters of an arbitrary python script of this repo, one rocprofv3 --pmc pass per counter group:
#   tools/pmc_any.sh <tag> "<script and args, relative to the repo root>" "<counters pass 1>" "<counters pass 2>" ...
# Output under gpurun_out/pmc_<tag>_<i>/ (+ gpurun_out/pmc_<tag>.json: per-kernel means, tools/pmc_summary.py).
tag=$1; cmd=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
i=0; dirs=""
for group in "$@"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 $R/$cmd > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i ($group): rc=$?"
  dirs="$dirs $R/gpurun_out/pmc_${tag}_$i"
done
python3 $R/tools/pmc_summary.py $dirs > $R/gpurun_out/pmc_$tag.json
