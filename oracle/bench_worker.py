"""One worker of bench.py's all-cores CPU baseline: loads the CPU oracle on a plain dump and answers a
shard of type-6 regions for a bounded time.  TEST/BENCH INFRASTRUCTURE ONLY (see oracle/vs_oracle.cpp).

usage: python -m oracle.bench_worker <plain dump> <regions.npy> <first> <last> <budget seconds>
prints one JSON line {"done": n, "seconds": t, "variants": v}
"""
import json
import sys
import time

import numpy as np

from oracle.oracle import Oracle


def main():
    plain, regions_path, lo, hi, budget = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    regions = np.load(regions_path)[lo:hi]
    orc = Oracle(plain)
    print("ready", flush=True)
    sys.stdin.readline()          # all workers start together, after every index is loaded
    done = nvar = 0
    t0 = time.perf_counter()
    for x, y in regions:
        n, _, _ = orc.get_var_in_ref(int(x), int(y), text=False)
        nvar += max(n, 0)
        done += 1
        if done >= 20 and time.perf_counter() - t0 > budget:
            break
    print(json.dumps({"done": done, "seconds": time.perf_counter() - t0, "variants": nvar}), flush=True)


if __name__ == "__main__":
    main()
