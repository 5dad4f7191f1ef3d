#!/bin/bash
# round 5, call 34: N = 384 in one workgroup for the 12-step blocks (shipped build: tests + bench)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c34; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_heads.py tests/test_gpu_persist.py tests/test_gpu_graph.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print('shipped', d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'])"
