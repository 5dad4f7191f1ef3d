#!/bin/bash
# Dev tool (developer build): the headline bench (3 batches in flight) under several environment settings: value, ms/step, one-batch rate
run() { echo "== $*"; env "$@" python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('one_batch_in_flight_images_per_sec'))"; }
for s in "$@"; do run $s; done
