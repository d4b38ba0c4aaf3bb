#!/bin/bash
# r06 call 2: whole -m gpu suite on the working tree (context-parallel tests included), then smoke
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
echo "== gpu tests"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -x > gpurun_out/r06_c02_tests.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r06_c02_tests.log
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
cp gpurun_out/test_metrics.jsonl gpurun_out/r06_c02_test_metrics.jsonl 2>/dev/null
echo "== done"
