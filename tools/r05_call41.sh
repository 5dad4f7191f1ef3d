#!/bin/bash
# round 5, call 41: ring depth 2 everywhere (32 KB per workgroup: all 1120 workgroups of the 3x3 head conv resident at once), developer build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c41; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['roofline']['sum_kernels_us'], [ (k.split(':',1)[1][2:34], round(v,1)) for k,v in list(d['roofline']['per_kernel_us'].items())[13:21]])"; }
run base A=1
run ring2 YK_X_NS3=1000
run r3mid YK_X_NS3HI=32
run base2 A=1
run r3mid2 YK_X_NS3HI=32
