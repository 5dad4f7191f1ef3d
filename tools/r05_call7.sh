#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c7; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for b in 32 64; do for il in 0 1; do
( YK_PIPE_IL=$il timeout 300 python tools/r05_igemm_sweep.py $b "/,14/2,11/3,15/2" ) 2>&1 | grep -v amdgpu.ids | sed "s/^/il=$il /"
done; done > $O/sweep.txt 2>&1
cat $O/sweep.txt
for il in 0 1; do YK_FORCE_MINK=64 YK_SPLIT_FORCE=1 YK_IGEMM_FORCE=15 YK_NS=2 YK_PIPE_IL=$il timeout 100 python tools/r05_igemm_phase.py 52 52 128 256 32 2>&1 | grep -v amdgpu.ids; done > $O/phase256.txt 2>&1; cat $O/phase256.txt
