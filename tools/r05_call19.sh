#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c19; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py tests/test_gpu_loss.py -x -q -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed\|Error" $O/tests.log | tail -4
( timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train.json 2> $O/train.err; python -c "import json;d=json.load(open('$O/train.json'));print(d['ms_per_step'], d['value'])"; tail -2 $O/train.err
