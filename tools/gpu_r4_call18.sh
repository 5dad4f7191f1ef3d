#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c18
O=gpurun_out/c18
export TMPDIR=/tmp
for cfg in "0 1" "1 1" "0 0"; do
set -- $cfg
( YK_PERSIST=$1 YK_HEADS=$2 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_p$1h$2.json 2> $O/bench_p$1h$2.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c18/bench_p$1h$2.json'))
print('persist=$1 heads=$2', {k:d[k] for k in ('value','value_from_host','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['config']['host_us_per_step'], d['roofline']['sum_kernels_us'])
PY
done
