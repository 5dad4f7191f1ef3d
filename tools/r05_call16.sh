#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c16; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for b in 32 64; do
( timeout 300 python tools/r05_igemm_sweep.py $b "/,14/2,16/3,17/3,17/4,18/3" ) 2>&1 | grep -v amdgpu.ids
done > $O/sweep.txt 2>&1
cat $O/sweep.txt
