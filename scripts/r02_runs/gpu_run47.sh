#!/bin/bash
# after compiling the attention experiments out of the product library: attention op tests + smoke
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attn or attention" 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
