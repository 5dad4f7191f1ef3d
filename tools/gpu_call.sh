#!/bin/bash
# The one GPU-box runner (through gpurun):  gpurun -- 'bash tools/gpu_call.sh <call-name>'
# sources tools/calls/<call-name>.sh (scratch, git-ignored) with these helpers defined; everything lands in gpurun_out/<call-name>/.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; CALL=$1; O=$R/gpurun_out/$CALL; mkdir -p "$O"
CS=$R/k210_yolo_framework_amd/csrc
# bench <tag> [ENV=..]...: one bench.py line (no CPU leg, no secondaries) -> $O/bench_<tag>.json + a one-line summary
bench() { local tag=$1; shift; ( env "$@" timeout 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open('$O/bench_$tag.json'))
    print('$tag', 'value', d['value'], 'one', d['config']['one_batch_in_flight_images_per_sec'], 'sum_us', d['roofline']['sum_kernels_us'])
except Exception as e:
    print('$tag', 'FAILED', e); print(open('$O/bench_$tag.err').read()[-1500:])
PY
}
# pmc <tag> "<counters>" <cmd...>: one --pmc pass (+ kernel trace, allowed beside --pmc) from /tmp
pmc() { local tag=$1 ctr=$2; shift 2; ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$tag -o p -- "$@" ) > $O/$tag.log 2>&1; echo "pmc $tag rc=$?"; }
# ktrace <tag> <cmd...>: kernel trace + stats
ktrace() { local tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o p -- "$@" ) > $O/$tag.log 2>&1; echo "ktrace $tag rc=$?"; }
gputests() { ( timeout ${2:-1500} python -m pytest $1 -x -q -m gpu ) > $O/tests.log 2>&1; tail -4 $O/tests.log; }
source $R/tools/calls/$CALL.sh
