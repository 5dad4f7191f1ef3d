#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c13
O=gpurun_out/c13
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_heads.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -25 $O/tests.log | cut -c1-300
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | tail -18
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 300 python tools/xheads_phase.py ) > $O/phases.log 2>&1
tail -6 $O/phases.log | cut -c1-200
