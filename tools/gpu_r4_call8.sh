#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
O=gpurun_out/c8
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -x -q 2>&1 | tail -15 ) > $O/tests.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for d in 0; do
  echo "=== YK_XP_DBG=$d" >> $O/phase.log
  ( YK_XP_DBG=$d timeout 120 python tools/xpersist_phase.py 2>&1 | grep -E "span|phase" ) >> $O/phase.log 2>&1
done
tail -15 $O/tests.log; cat $O/phase.log
