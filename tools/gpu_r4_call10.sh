#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c10
O=gpurun_out/c10
export TMPDIR=/tmp
( timeout 600 python tools/xbench.py ) > $O/xbench.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err )
cut -c1-110 $O/xbench.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c10/bench.json'))
print({k:d[k] for k in ('value','value_from_host','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['config']['host_us_per_step'], d['roofline']['sum_kernels_us'])
print(d['secondary'])
PY
tail -3 $O/bench.err
