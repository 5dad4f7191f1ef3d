#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c29
O=gpurun_out/c29
export TMPDIR=/tmp
for m in side eager copy; do
( YK_H2D=$m timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c29/bench_$m.json'))
s=d['secondary']
print('$m', {k:d[k] for k in ('value','value_from_host','from_host_frac_of_value')}, d['config']['from_host_host_us_per_step'], 'eager_from_host', s.get('eager_from_host_images_per_sec'), 'lb_from_host', s.get('from_host_letterbox_images_per_sec'))
PY
tail -1 $O/bench_$m.err | cut -c1-200
done
