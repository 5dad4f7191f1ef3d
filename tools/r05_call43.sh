#!/bin/bash
# round 5, call 43: A/B on ONE box (boxes differ by 3 %): ring depth of the K-split launches x staging-free LDS of the direct-store blocks
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c43; mkdir -p $O
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['roofline']['sum_kernels_us'])"; }
for r in 1 2; do
run old_$r YK_X_SK_NS3=1 YK_XB_NOSHRINK=1
run ring2_$r YK_XB_NOSHRINK=1
run shrink_$r YK_X_SK_NS3=1
run both_$r A=1
done
