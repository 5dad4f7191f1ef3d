#!/bin/bash
# round 5, call 21: four workgroups per CU for every fused block whose tile allows 128 registers (launch bounds only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c21; mkdir -p $O
( timeout 300 python tools/xbench.py yolo_mobilev1 32 ) > $O/xbench.log 2>&1; sed -n 1,13p $O/xbench.log | cut -c1-110
( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us']); print({k[:40]:v for k,v in list(d['roofline']['per_kernel_us'].items())[:8]})"
