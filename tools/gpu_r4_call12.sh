#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c12
O=gpurun_out/c12
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_graph.py tests/test_gpu_persist.py "tests/test_gpu_net.py::test_f16x2_whole_network_undamped_vs_fp32_oracle" -m gpu -q ) > $O/tests.log 2>&1
tail -8 $O/tests.log
( timeout 900 bash tools/run_asan.sh ) > $O/asan.log 2>&1
grep -v "^  File" $O/asan.log | tail -12
