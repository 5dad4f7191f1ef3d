#!/bin/bash
# round 5, call 25: the 384-channel blocks at 14x20 as fused launches again (weight tile out of LDS), vs the two-launch form; xdw at 128 registers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c25; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for nk in 6 12; do
  ( YK_XB_MAXNK=$nk timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_nk$nk.json 2> $O/bench_nk$nk.err; python -c "
import json; d=json.load(open('$O/bench_nk$nk.json')); print($nk, d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us']); print([ (k.split(':',1)[1][:34], round(v,1)) for k,v in list(d['roofline']['per_kernel_us'].items())[6:20]])"
done
