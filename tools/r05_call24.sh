#!/bin/bash
# round 5, call 24: two-stage fused blocks again, now that the weight tile is out of LDS (developer build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c24; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XBS_DB=1 XBS_TN=2,3 timeout 900 python tools/xbsweep.py 1 5 ) > $O/sweep_db1.log 2>&1
grep "^block" $O/sweep_db1.log | cut -c1-600
