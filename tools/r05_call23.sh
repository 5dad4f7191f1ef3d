#!/bin/bash
# round 5, call 23: fused-block geometry sweep with the weight tile out of LDS (developer build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c23; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XBS_DB=0 XBS_TN=1 timeout 400 python tools/xbsweep.py 0 0 ) > $O/sweep0.log 2>&1
( XBS_DB=0 XBS_TN=2,3 timeout 900 python tools/xbsweep.py 1 5 ) > $O/sweep1.log 2>&1
cat $O/sweep0.log $O/sweep1.log | grep "^block" | cut -c1-700
