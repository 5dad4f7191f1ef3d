#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c26
O=gpurun_out/c26
export TMPDIR=/tmp
for v in 1 0; do
( HSA_ENABLE_SDMA=$v timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_sdma$v.json 2> $O/bench_sdma$v.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c26/bench_sdma$v.json'))
s=d['secondary']
print('HSA_ENABLE_SDMA=$v', {k:d[k] for k in ('value','value_from_host')}, 'eager_from_host', s.get('eager_from_host_images_per_sec'), 'lb_from_host', s.get('from_host_letterbox_images_per_sec'))
PY
done
