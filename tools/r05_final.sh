#!/bin/bash
# round-5 final measurements (GPU box): the bench line, the other BASELINE configurations, the smoke entry
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final; mkdir -p $O
( timeout 900 python bench.py ) > $O/bench_line.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_line.json')); print({k:d[k] for k in ('value','value_from_host','from_host_frac_of_value','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['roofline']['frac'], d['roofline']['avg_us'], d['cpu_baseline']['value'], d['secondary'].get('train'))"
( timeout 600 python tools/netbench.py ) > $O/other_configs.txt 2>&1; grep "^|" $O/other_configs.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
