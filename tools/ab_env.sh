#!/bin/bash
# Same-box A/B of an environment switch of the engine on the default bench:   tools/ab_env.sh VAR "v1 v2 ..." [rounds] [extra bench args]
# (box to box -- and run to run on one box -- the headline moves by several per cent: settings alternate, several rounds)
var=$1; vals=$2; rounds=${3:-2}; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/ab_env
for r in $(seq 1 $rounds); do
  for v in $vals; do
    env $var=$v timeout 400 python3 bench.py --no-cpu-baseline --latency-samples 0 --extras none "$@" > gpurun_out/ab_env/${var}_${v}_$r.json 2>/dev/null
    python3 - gpurun_out/ab_env/${var}_${v}_$r.json "$var=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "%.1f M regions/s  step %.4f ms  expansion in the loop %.4f / alone %.4f ms" % (d["value"] / 1e6, d["ms_per_step"], r.get("avg_launch_ms") or 0, r.get("avg_launch_ms_alone") or 0), flush=True)
PY
  done
done
