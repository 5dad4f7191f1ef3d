#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c22
O=gpurun_out/c22
export TMPDIR=/tmp
for s in 3 4 5 6 8; do
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --streams $s > $O/bench_s$s.json 2> $O/bench_s$s.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c22/bench_s$s.json'))
print('streams=$s', {k:d[k] for k in ('value','ms_per_step')}, d['config']['host_us_per_step'])
PY
done
