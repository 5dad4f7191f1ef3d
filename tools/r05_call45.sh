#!/bin/bash
# round 5, call 45: the H2D leg of submit_host on the pipeline's own copy stream, two device input buffers per slot
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c45; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
( timeout 300 python bench.py --from-host --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_fh$i.json 2> $O/bench_fh$i.err; python -c "
import json; d=json.load(open('$O/bench_fh$i.json')); print('from host', d['value'], d['config']['one_batch_in_flight_images_per_sec'])"
( timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_r$i.json 2> $O/bench_r$i.err; python -c "
import json; d=json.load(open('$O/bench_r$i.json')); print('resident', d['value'], d['config']['one_batch_in_flight_images_per_sec'])"
done
