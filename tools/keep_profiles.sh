#!/bin/bash
# After a tools/profile_round.sh run came back through gpurun: copy what is judged from gpurun_out/profiles_<tag>/ into profiles/ --
# the round's files, the workload's traffic file (bench.py attaches counter traffic only when its kernel hash is this tree's: a stale
# one silently turns roofline.traffic into null) and the kernel hash itself.   tools/keep_profiles.sh <tag>
tag=$1
src=gpurun_out/profiles_$tag
[ -d "$src" ] || { echo "no $src"; exit 1; }
cp $src/${tag}_* profiles/
cp $src/traffic_*.json profiles/
cp $src/kernels_blob.txt profiles/kernels_blob.txt
python3 - <<PY
import sys, json, glob
sys.path.insert(0, ".")
import bench
h = bench.kernels_hash()
for f in sorted(glob.glob("profiles/traffic_*.json")):
    b = json.load(open(f)).get("kernels_blob")
    print(f, "ok" if b == h else "STALE (%s, tree %s)" % (b, h))
PY
