#!/bin/bash
# round 5, call 22: pointwise weight fragments straight into registers (no LDS tile)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c22; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout 300 python tools/xbench.py yolo_mobilev1 32 ) > $O/xbench.log 2>&1; grep -n "stem\|sum\|err" $O/xbench.log | head
( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['frac'], d['roofline']['avg_us'])"
