#!/bin/bash
# round 5, call 48: A/B on one box, previous commit's library vs the stem block specialised on the frame type
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c49; mkdir -p $O
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'], [round(v,1) for v in list(d['roofline']['per_kernel_us'].values())[1:3]])"; }
P=$PWD/k210_yolo_framework_amd/csrc
for r in 1 2; do
run prev_$r YK_LIB_PATH=$P/libyolo_hip_prev.so
run new_$r YK_LIB_PATH=$P/libyolo_hip.so
done
