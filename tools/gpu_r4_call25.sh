#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c25
O=gpurun_out/c25
export TMPDIR=/tmp
( timeout 300 python tools/xbench.py ) > $O/xbench.log 2>&1
cut -c1-150 $O/xbench.log | sed -n 4,12p
( timeout 900 bash tools/run_asan.sh ) > $O/asan.log 2>&1
grep -v "^  File" $O/asan.log | grep -n "passed\|failed\|runtime error\|== " | head -8 | cut -c1-220
( timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c25/bench.json'))
print({k:d[k] for k in ('value','value_from_host','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'], d['roofline']['avg_us'])
PY
