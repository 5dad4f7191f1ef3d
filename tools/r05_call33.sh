#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c33; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
for v in "6 6" "6 0"; do set -- $v
( YK_XB_TNL=$1 YK_XB_TNL_NK=$2 YK_XB_TML=4 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; python -c "
import json; d=json.load(open('$O/bench_$1_$2.json')); print('tn', $1, 'nk>', $2, d['value'], d['roofline']['sum_kernels_us'], [ (k.split(':',1)[1][22:50], round(v,1)) for k,v in list(d['roofline']['per_kernel_us'].items())[6:8]])"
done
