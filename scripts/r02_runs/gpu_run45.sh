#!/bin/bash
# final sanity of the round: full GPU suite + smoke on the library as committed
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=600 > gpurun_out/tests_final.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
