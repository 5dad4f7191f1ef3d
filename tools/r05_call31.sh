#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c31; mkdir -p $O
for i in 1 2; do
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us']); print([ (k.split(':',1)[1][:40], round(v,1)) for k,v in list(d['roofline']['per_kernel_us'].items())[6:9]])"
done
