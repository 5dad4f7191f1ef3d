#!/bin/bash
# round 5, call 28: phase stamps of the fused blocks in the throughput plan (developer build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c28; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( XPH_SCHEDULE=throughput timeout 300 python tools/xphase.py 3,5,7 ) > $O/phases.log 2>&1
cat $O/phases.log | grep -v amdgpu.ids
