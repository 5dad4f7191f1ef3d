#!/bin/bash
# round 5, call 27: geometry / stages of the fused 14x20x384 block (developer build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c27; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XBS_DB=0,1 XBS_TN=2,3,6 timeout 900 python tools/xbsweep.py 6 6 ) > $O/sweep384.log 2>&1
grep "^block" $O/sweep384.log | cut -c1-900
