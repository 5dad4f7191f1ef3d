#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c19
O=gpurun_out/c19
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -6 $O/tests.log | cut -c1-300
( timeout 600 python bench.py --mode train --steps 30 --warmup 5 > $O/train.json 2> $O/train.err ); tail -c 600 $O/train.json; tail -3 $O/train.err
