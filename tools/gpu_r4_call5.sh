#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
O=gpurun_out/c5
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -x -q 2>&1 | tail -40 ) > $O/tests.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
unset YK_LIB_PATH
tail -40 $O/tests.log; cat $O/xbench.log | cut -c1-120
