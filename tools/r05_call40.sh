#!/bin/bash
# round 5, call 40: mixed schedules with four batches in flight (heads as the cluster launch / late backbone as the persistent launch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c40; mkdir -p $O
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['roofline']['sum_kernels_us'], d['config']['launches_per_step'])"; }
run base A=1
run heads YK_HEADS=1
run persist YK_PERSIST=1
run both YK_HEADS=1 YK_PERSIST=1
