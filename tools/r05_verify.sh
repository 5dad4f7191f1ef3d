#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5verify; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ) > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print(d['value'], d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['frac'])"
