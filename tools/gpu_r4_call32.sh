#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c32
O=gpurun_out/c32
export TMPDIR=/tmp
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | sed -n 2,14p
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_kmodel.py tests/test_gpu_heads.py tests/test_gpu_persist.py tests/test_gpu_map.py -m gpu -x -q ) > $O/tests.log 2>&1
grep -n "passed\|failed" $O/tests.log | tail -3
