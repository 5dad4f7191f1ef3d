#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c13; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -3
( timeout 400 python tools/netbench.py yolo ) > $O/netbench_yolo.txt 2>&1; grep "^|" $O/netbench_yolo.txt
( timeout 200 python tools/darknet_layers.py f16 64 ) > $O/darknet_f16_b64.txt 2>&1; head -3 $O/darknet_f16_b64.txt | tail -2
( timeout 700 python bench.py ) > $O/bench_line.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_line.json')); print({k:d[k] for k in ('value','value_from_host','from_host_frac_of_value','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['secondary'].get('train'))"
