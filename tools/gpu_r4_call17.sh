#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c17
O=gpurun_out/c17
export TMPDIR=/tmp
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | tail -3
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 300 python tools/xheads_phase.py ) > $O/phases.log 2>&1
tail -4 $O/phases.log | cut -c1-260
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
tail -8 $O/tests.log | cut -c1-300
