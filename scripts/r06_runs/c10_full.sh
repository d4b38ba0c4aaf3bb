#!/bin/bash
# r06 call 10: the whole -m gpu suite at the closing tree + smoke; metrics kept for profiles/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/test_metrics.jsonl
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/r06_c10_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_c10_tests.log
cp gpurun_out/test_metrics.jsonl gpurun_out/r06_test_metrics.jsonl
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== done"
