#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
O=gpurun_out/c4
export TMPDIR=/tmp
( timeout 300 tools/dma_probe.bin ) > $O/dma_probe.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev_base.so
( timeout 900 python tools/xsweep.py "YK_X_DBG=16" "YK_X_DBG=1" "YK_X_DBG=5" "YK_X_DBG=20" "YK_X_SPLITK=1 YK_X_DBG=16" "YK_X_SPLITK=1 YK_X_NS=4" "YK_X_SPLITK=1 YK_X_NS=4 YK_X_DBG=16" ) > $O/xsweep.log 2>&1
cat $O/dma_probe.log; cut -c1-250 $O/xsweep.log
