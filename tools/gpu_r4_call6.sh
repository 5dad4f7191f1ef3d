#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
O=gpurun_out/c6
export TMPDIR=/tmp
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( timeout 300 python tools/xpersist_phase.py ) > $O/phase.log 2>&1
cat $O/phase.log
