#!/bin/bash
# r06 call 13: 2-rank dry run of bench.py --gpus 2 on ONE GPU (gloo: the ranks share the box's GPU) -- the N > 1 line with the context-parallel stream leg
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out
M3R_DIST_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --scenes 4 --stream-frames 40 > $O/r06_bench_2rank_gloo_dryrun.log 2>&1; echo "rc=$?"
tail -c 3000 $O/r06_bench_2rank_gloo_dryrun.log | python -c "
import sys,json
t=sys.stdin.read()
i=t.rfind('{\"metric\"')
try:
    d=json.loads(t[i:].strip().splitlines()[0])
    print('value',d['value'],'n_gpus',d['n_gpus'],'rccl',d.get('rccl'))
    print('view_sharded',json.dumps(d.get('view_sharded'))[:1500])
except Exception as e:
    print('parse failed',e); print(t[-1500:])
"
