#!/bin/bash
# r04 GPU call 6: 2-rank gloo dry run of `bench.py --gpus 2` (both ranks on the box's one GPU): the N > 1 code paths of bench.py and parallel.py
# (replicas + weak / strong view-sharded scene + sharded stream) with the single-collective keyframe gather of r04
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
M3R_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --scenes 4 --stream-frames 40 > $O/r04_bench_2rank_gloo_dryrun.log 2>&1; echo "rc=$?"
tail -1 $O/r04_bench_2rank_gloo_dryrun.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','value_single_scene','n_gpus','ms_per_step','scaling')}, d['config']['parallelism'])
for c in d['configs']: print(c['config'][:110], c['value'], c['unit'])
print(d['multi_gpu'])"
echo "== done"
