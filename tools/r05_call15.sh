#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c15; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_heads.py tests/test_gpu_e2e.py tests/test_gpu_graph.py -q -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -3
( YK_CLUSTER_WT=1 timeout 300 python -m pytest tests/test_gpu_persist.py tests/test_gpu_heads.py -q -m gpu ) > $O/tests_wt.log 2>&1; grep -n "passed\|failed" $O/tests_wt.log | tail -3
( timeout 120 python tools/r05_rate.py latency ) 2>&1 | grep -v amdgpu | tail -1
