#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c10; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
run() { echo -n "$1 :: "; env $1 timeout 120 python tools/r05_rate.py 2>&1 | grep -v amdgpu.ids | tail -1; }
{
run "YK_X_IL=0"
run "YK_X_IL=1"
run "YK_X_IL=0 YK_XB_ALWAYS=1"
run "YK_X_IL=0 YK_X_NS=3"
run "YK_X_IL=0 YK_SPLITK=0"
run "YK_X_IL=0 YK_X_CFG=0"
run "YK_X_IL=0 YK_X_CFG=1"
run "YK_X_IL=0 YK_PERSIST=1"
run "YK_X_IL=0 YK_HEADS=1"
run "YK_X_IL=0 YK_X_DWGS=6"
run "YK_X_IL=0 YK_X_DWLDS=24"
} > $O/rates.txt 2>&1
cat $O/rates.txt
