#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c21
O=gpurun_out/c21
export TMPDIR=/tmp
( timeout 600 python bench.py --mode train --steps 30 --warmup 5 > $O/train_new.json 2> $O/train_new.err ); cut -c1-330 $O/train_new.json
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q ) > $O/tests.log 2>&1
grep -n "passed\|failed" $O/tests.log | tail -2
