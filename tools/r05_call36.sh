#!/bin/bash
# round 5, call 36: batches in flight and split-K again, with the round's fused blocks (shipped build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c36; mkdir -p $O
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $EXTRA ) > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['roofline']['sum_kernels_us'])"; }
EXTRA="--streams 3" run s3 A=1
EXTRA="--streams 4" run s4 A=1
EXTRA="--streams 5" run s5 A=1
EXTRA="--streams 4" run nosplitk YK_SPLITK=0
EXTRA="--streams 4 --batch 64" run b64 A=1
EXTRA="--streams 2 --batch 64" run b64s2 A=1
