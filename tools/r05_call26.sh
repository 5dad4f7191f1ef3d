#!/bin/bash
# round 5, call 26: fused 384-channel blocks in the throughput schedule (shipped build: tests), the 7x10 blocks fused too (developer build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c26; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_heads.py tests/test_gpu_persist.py tests/test_gpu_graph.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print('shipped', d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'])"
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for v in "12 128" "24 64" "12 64"; do set -- $v
  ( YK_XB_MAXNK=$1 YK_XB_MINPX=$2 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ) > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; python -c "
import json; d=json.load(open('$O/bench_$1_$2.json')); print('dev', $1, $2, d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us']); print([ (k.split(':',1)[1][:30], round(v,1)) for k,v in list(d['roofline']['per_kernel_us'].items())[11:18]])"
done
