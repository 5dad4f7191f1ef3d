#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c11; mkdir -p $O
( YK_TRAIN_WSTREAM=4 timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for n in 0 1 2 4 8; do
( YK_TRAIN_WSTREAM=$n timeout 200 python bench.py --mode train --steps 30 --warmup 3 ) > $O/train_w$n.json 2> $O/train_w$n.err; echo -n "WSTREAM=$n "; python -c "import json;d=json.load(open('$O/train_w$n.json'));print(d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
