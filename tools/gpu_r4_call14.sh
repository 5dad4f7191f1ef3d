#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c14
O=gpurun_out/c14
export TMPDIR=/tmp
for h in 1 0; do
( YK_HEADS=$h timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_h$h.json 2> $O/bench_h$h.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c14/bench_h$h.json'))
print('heads=$h', {k:d[k] for k in ('value','value_from_host','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['config']['host_us_per_step'], d['roofline']['sum_kernels_us'])
PY
done
