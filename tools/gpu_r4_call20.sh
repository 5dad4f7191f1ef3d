#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c20
O=gpurun_out/c20
export TMPDIR=/tmp
( YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_oldtrain.so timeout 600 python bench.py --mode train --steps 30 --warmup 5 > $O/train_old.json 2> $O/train_old.err ); cut -c1-330 $O/train_old.json
( timeout 600 python bench.py --mode train --steps 30 --warmup 5 > $O/train_new.json 2> $O/train_new.err ); cut -c1-330 $O/train_new.json
( timeout 600 python bench.py --mode train --steps 30 --warmup 5 > $O/train_new2.json 2> $O/train_new2.err ); cut -c1-330 $O/train_new2.json
