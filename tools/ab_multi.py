#!/usr/bin/env python3
"""Same-box comparison of several engine builds: python tools/ab_multi.py name=path.so ... (the in-tree library is 'tree')."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [("tree", None)] + [tuple(a.split("=", 1)) for a in sys.argv[1:]]
for _ in range(2):
    for name, lib in libs:
        env = dict(os.environ, VS_BENCH_SKIP_T4="1")
        if lib:
            env["VS_ENGINE_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                              "--latency-samples", "50"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(name, "FAILED", out.stderr[-400:])
            continue
        print(f"{name:>10}: {d['value'] / 1e6:.2f} M q/s  fill {d['roofline']['avg_launch_ms']:.3f} ms  frac {d['roofline']['frac']:.3f}  "
              f"p50 {d['p50_latency_us']:.1f} us  digest {d['result_digest']}", flush=True)
