#!/usr/bin/env python3
"""Time k_fill_carriers with one density regime skipped at a time (VS_FILL_ABLATE bits: 1 sparse, 2 medium,
4 dense) -- profiling aid; results with regimes skipped are of course wrong."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for a in [int(x) for x in os.environ.get('ABLATE_LIST', '0,1,2,4,6,5,3,7').split(',')]:
    env = dict(os.environ, VS_FILL_ABLATE=str(a), VS_BENCH_SKIP_T4="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                          "--latency-samples", "10"], env=env, capture_output=True, text=True)
    line = out.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    print(f"ablate={a} (skip{' sparse' if a & 1 else ''}{' medium' if a & 2 else ''}{' dense' if a & 4 else ''}): "
          f"fill {d['roofline']['avg_launch_ms']:.3f} ms", flush=True)
