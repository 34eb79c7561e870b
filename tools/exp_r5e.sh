#!/bin/bash
# A/B of engine builds (VS_ENGINE_LIB) on the walking query types' legs of bench.py: LIBS="old cur", WLS="chr22-100 chr1-2504"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp_r5e
mkdir -p $OUT
cd $R
for wl in ${WLS:-chr22-100 chr1-2504}; do
for rep in 1 2; do
for lib in ${LIBS:-old cur}; do
  export VS_ENGINE_LIB=$R/variantstore_amd/lib/ab/$lib.so
  timeout 900 python3 bench.py --workload $wl --extras t4,sc --no-cpu-baseline --latency-samples 0 > $OUT/$lib.$wl.$rep.json 2> $OUT/$lib.$wl.$rep.err
  python3 - $OUT/$lib.$wl.$rep.json $lib $wl <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
t4, sc = d.get("type4") or {}, d.get("sample_coordinate_queries") or {}
print(sys.argv[2], sys.argv[3], "type 6 %.1f M/s |" % (d["value"] / 1e6), "type 4 %.1f M/s (%.4f ms, walk %.4f) |" % (t4["queries_per_s"] / 1e6, t4["ms_per_batch"], t4["walk_phase_ms"]),
      "types 2 / 3 / 5: %.1f / %.1f / %.1f M/s" % tuple(sc["type%d_queries_per_s" % k] / 1e6 for k in (2, 3, 5)))
PY
done
done
done
unset VS_ENGINE_LIB
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r5e -o p -- python3 $R/bench.py --workload chr22-100 --steps 6 --warmup 2 --no-cpu-baseline --latency-samples 0 --extras t4 > $OUT/trace.log 2>&1
cd $R
python3 tools/trace_walk.py $(ls gpurun_out/prof_r5e/*.db | head -1) | tee $OUT/walk_trace.txt
