#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c2; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( YK_TRAIN_WSTREAM=0 timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train_w0.json 2> $O/train_w0.err; cut -c1-300 $O/train_w0.json
( YK_TRAIN_WSTREAM=1 timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train_w1.json 2> $O/train_w1.err; cut -c1-300 $O/train_w1.json
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout 400 python tools/r05_igemm_sweep.py 32 ) > $O/igemm_sweep_b32.txt 2>&1; cat $O/igemm_sweep_b32.txt
