#!/bin/bash
# Rehearsal of bench.py's N = 2 code on ONE GPU: torch.distributed over gloo, the engine's collective over tests/native/fake_rccl.cpp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rehearsal
mkdir -p $OUT
cd $R
LIB=$(python3 -c "import sys; sys.path.insert(0,'tests'); from helpers import build_fake_rccl; print(build_fake_rccl())")
export VS_RCCL_LIB=$LIB VS_BENCH_SAME_DEVICE=1 VS_FAKE_RCCL_TIMEOUT_S=300 HSA_ENABLE_IPC_MODE_LEGACY=0
WL=${1:-chr22-100}
timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 2 --workload $WL --no-cpu-baseline --latency-samples 0 > $OUT/n2_$WL.json 2> $OUT/n2_$WL.err
echo "rc=$?"; tail -c 1500 $OUT/n2_$WL.json; tail -5 $OUT/n2_$WL.err
rm -f /dev/shm/vs_fake_rccl_*
