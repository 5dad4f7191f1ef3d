#!/bin/bash
# round 5, call 47: the fused stem block specialised on the frame type (u8 | fp32) at compile time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c47; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_heads.py tests/test_gpu_persist.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'], [ round(v,1) for k,v in list(d['roofline']['per_kernel_us'].items())[1:8]])"
done
