#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c6; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so YK_FORCE_MINK=64 YK_SPLIT_FORCE=1
for v in "9 2 0" "9 2 1" "14 2 0" "14 2 1" "14 3 0" "14 3 1" "11 3 0" "11 3 1"; do
  set -- $v
  YK_IGEMM_FORCE=$1 YK_NS=$2 YK_PIPE_IL=$3 timeout 100 python tools/r05_igemm_phase.py 52 52 128 256 32 2>&1 | grep -v amdgpu.ids
done > $O/phase.txt 2>&1
cat $O/phase.txt
