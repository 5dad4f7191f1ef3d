#!/bin/bash
# round 5, call 35: what the depthwise pass's LDS reads cost with four batches in flight.  Pricing builds (-DYK_KO: results are wrong on
# purpose): ko1 = no weight reads (18 of the 40 b128 reads per item left), ko3 = no weight reads and one patch read per tap ROW (6 left)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c35; mkdir -p $O
for v in dev ko1 ko3; do
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_$v.so
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'], [ round(v,1) for k,v in list(d['roofline']['per_kernel_us'].items())[1:8]])"
done
