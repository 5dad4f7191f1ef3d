#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c30
O=gpurun_out/c30
export TMPDIR=/tmp
( YK_BENCH_FH_TWICE=1 timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c30/bench.json'))
s=d['secondary']
print({k:d[k] for k in ('value','value_from_host','from_host_frac_of_value')}, 'eager_from_host', s.get('eager_from_host_images_per_sec'), 'lb_from_host', s.get('from_host_letterbox_images_per_sec'))
PY
grep "from-host again" $O/bench.err
