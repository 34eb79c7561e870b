#!/usr/bin/env python3
"""Same-box A/B of two engine builds: python tools/ab_bench.py <base .so> [rounds]   (box-to-box variance is +-5 %,
so two builds are only comparable when they alternate on one box).  The library under test is the in-tree one."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for _ in range(rounds):
    for name, lib in (("base", base), ("new ", None)):
        env = dict(os.environ, VS_BENCH_SKIP_T4="1")
        if lib:
            env["VS_ENGINE_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                              "--latency-samples", "50"], env=env, capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name}: {d['value'] / 1e6:.2f} M q/s  fill {d['roofline']['avg_launch_ms']:.3f} ms  frac {d['roofline']['frac']:.3f}  "
              f"p50 {d['p50_latency_us']:.1f} us  digest {d['result_digest']}", flush=True)
