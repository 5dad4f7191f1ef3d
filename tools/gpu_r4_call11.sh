#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c11
O=gpurun_out/c11
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_graph.py tests/test_gpu_pipeline.py -m gpu -x -q ) > $O/new_tests.log 2>&1
tail -15 $O/new_tests.log
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
tail -8 $O/tests.log
( timeout 600 bash tools/run_asan.sh ) > $O/asan.log 2>&1
tail -6 $O/asan.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err )
python - <<'PY'
import json
d=json.load(open('gpurun_out/c11/bench.json'))
print({k:d[k] for k in ('value','value_from_host','ms_per_step')}, d['config']['one_batch_in_flight_images_per_sec'], d['config']['host_us_per_step'], d['roofline']['sum_kernels_us'])
print(d['cpu_baseline'])
PY
tail -3 $O/bench.err
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
