#!/bin/bash
# N back-to-back runs of the headline loop on one box (each a fresh process and a fresh index): run-to-run spread of the
# bench line.  Usage (through gpurun, from the repo root): tools/bench_repeats.sh [N]
n=${1:-8}
for i in $(seq 1 $n); do
  python bench.py --extras none --no-cpu-baseline --latency-samples 0 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('run $i: value %.1f M/s  ms_per_step %.4f  expansion in the loop %.4f ms  alone %.4f ms  frac %.3f  frac_alone %.3f  digest %s'
      % (j['value'] / 1e6, j['ms_per_step'], r['avg_launch_ms'], r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], j['result_digest']))"
done
