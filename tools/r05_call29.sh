#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c29; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XBS_ALL=1 XBS_DB=0,1 XBS_TN=3,6 timeout 900 python tools/xbsweep.py 6 6 ) > $O/sweep384.log 2>&1
grep -v amdgpu.ids $O/sweep384.log | head -70
