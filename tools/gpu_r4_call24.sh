#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c24
O=gpurun_out/c24
export TMPDIR=/tmp
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | head -12
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_kmodel.py -m gpu -x -q ) > $O/tests.log 2>&1
grep -n "passed\|failed" $O/tests.log | tail -3
( timeout 900 bash tools/run_asan.sh ) > $O/asan.log 2>&1
grep -v "^  File" $O/asan.log | grep -n "passed\|failed\|runtime error\|== " | head -8 | cut -c1-220
