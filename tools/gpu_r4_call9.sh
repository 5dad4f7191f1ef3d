#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
O=gpurun_out/c9
rm -f $O/phase.log
export TMPDIR=/tmp
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for d in 0 1 2 3; do
  echo "=== YK_XP_DBG=$d" >> $O/phase.log
  ( YK_XP_DBG=$d timeout 120 python tools/xpersist_phase.py 2>&1 | grep -E "span|phase  [3456] " ) >> $O/phase.log 2>&1
done
cat $O/phase.log
