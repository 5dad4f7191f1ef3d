#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c27
O=gpurun_out/c27
export TMPDIR=/tmp
( timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c27/bench.json'))
s=d['secondary']
print({k:d[k] for k in ('value','value_from_host','from_host_frac_of_value')}, d['config']['one_batch_in_flight_images_per_sec'], d['config']['from_host_host_us_per_step'], 'eager_from_host', s.get('eager_from_host_images_per_sec'), 'lb_from_host', s.get('from_host_letterbox_images_per_sec'))
PY
tail -2 $O/bench.err
