#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c16
O=gpurun_out/c16
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_heads.py tests/test_gpu_persist.py tests/test_abi.py tests/test_gpu_graph.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -12 $O/tests.log | cut -c1-300
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | tail -4
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 300 python tools/xheads_phase.py ) > $O/phases.log 2>&1
tail -4 $O/phases.log | cut -c1-260
