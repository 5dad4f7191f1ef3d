#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c31
O=gpurun_out/c31
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_heads.py tests/test_gpu_persist.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -5 $O/tests.log | cut -c1-300
( YK_CLUSTER_WT=1 timeout 300 python tools/xbench.py ) > $O/xbench_wt.log 2>&1
cut -c1-150 $O/xbench_wt.log | tail -3
