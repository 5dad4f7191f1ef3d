#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c3; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py tests/test_gpu_train.py tests/test_gpu_train_dp.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for il in 0 1; do
  ( YK_X_IL=$il timeout 120 python tools/xbench.py yolo_mobilev1 32 ) > $O/xbench_il$il.txt 2>&1; tail -1 $O/xbench_il$il.txt
  ( YK_X_IL=$il YK_PERSIST=0 YK_HEADS=0 timeout 120 python tools/xbench.py yolo_mobilev1 32 ) > $O/xbench_thr_il$il.txt 2>&1; tail -1 $O/xbench_thr_il$il.txt
  ( YK_PIPE_IL=$il timeout 200 python tools/darknet_layers.py f16 32 ) > $O/darknet_f16_b32_il$il.txt 2>&1; head -3 $O/darknet_f16_b32_il$il.txt | tail -2
  ( YK_X_IL=$il timeout 200 python tools/darknet_layers.py f16x2 32 ) > $O/darknet_x2_b32_il$il.txt 2>&1; head -3 $O/darknet_x2_b32_il$il.txt | tail -2
done
( YK_TRAIN_WSTREAM=0 timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train_w0.json 2> $O/train_w0.err; cut -c1-250 $O/train_w0.json
( YK_TRAIN_WSTREAM=1 timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train_w1.json 2> $O/train_w1.err; cut -c1-250 $O/train_w1.json; tail -2 $O/train_w1.err
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 400 python tools/r05_igemm_sweep.py 32 ) > $O/igemm_sweep_b32_il.txt 2>&1; cat $O/igemm_sweep_b32_il.txt
