#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c5; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_layers.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for il in 0 1; do
  ( YK_PIPE_IL=$il timeout 200 python tools/darknet_layers.py f16 32 ) > $O/darknet_f16_b32_il$il.txt 2>&1; head -3 $O/darknet_f16_b32_il$il.txt | tail -2
done
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 400 python tools/r05_igemm_sweep.py 32 ) > $O/igemm_sweep_b32_il.txt 2>&1; cat $O/igemm_sweep_b32_il.txt
