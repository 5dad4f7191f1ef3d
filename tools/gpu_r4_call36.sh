#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c36
O=gpurun_out/c36
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1
grep -n "passed\|failed" $O/tests.log | tail -3
( timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err )
python - <<PY
import json
d=json.load(open('gpurun_out/c36/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['sum_kernels_us'], d['config']['latency_schedule']['sum_kernels_us'])
print(json.dumps(d['roofline']['families'])[:700])
PY
