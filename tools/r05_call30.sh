#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c30; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XPH_SCHEDULE=throughput YK_XB_LAYER=6 YK_XB_DB=1 YK_XB_TM=4 YK_XB_TN=3 YK_XB_TW=4 timeout 300 python tools/xphase.py 7 ) > $O/phases_db1.log 2>&1
grep -E "^x:|step 1|loop end|copied|landed" $O/phases_db1.log
