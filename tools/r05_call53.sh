#!/bin/bash
# round 5, call 53: staged epilogue of the fused blocks through per-image descriptors (output and residual): whole GPU suite + the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c53; mkdir -p $O
( timeout 1500 python -m pytest tests -q -x -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'])"
( timeout 300 python tools/netbench.py yolo_mobilev2 ) > $O/v2.txt 2>&1; grep "^| yolo_mobilev2" $O/v2.txt
