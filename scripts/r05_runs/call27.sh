#!/bin/bash
# r05 GPU call 27: 2-rank gloo dry run of `bench.py --gpus 2` at the final kernels (both ranks on the box's one GPU; what the driver launches with RCCL on a node)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
M3R_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --scenes 4 --stream-frames 40 > $O/r05_bench_2rank_gloo_dryrun.log 2>&1; echo "rc=$?"
tail -1 $O/r05_bench_2rank_gloo_dryrun.log | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('metric', 'value', 'n_gpus', 'scaling', 'rccl', 'view_sharded')})
"
echo "== done"
