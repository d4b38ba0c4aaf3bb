#!/bin/bash
# 2-rank dry run of the N > 1 benchmark path on a 1-GPU box (gloo backend, both ranks share the GPU): weak headline + strong + sharded streaming
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 M3R_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --stream-frames 40 > gpurun_out/bench_2rank_gloo.log 2>&1; echo "rc=$?"
tail -c 3000 gpurun_out/bench_2rank_gloo.log
