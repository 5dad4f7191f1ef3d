#!/bin/bash
# Run ON THE GPU BOX (through gpurun) at the end of a round: profile refresh (tools/refresh_profiles5.sh), the whole -m gpu suite,
# the sanitizer flavours (tools/run_asan.sh) and __graft_entry__.smoke().
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/round_end
O=gpurun_out/round_end
export TMPDIR=/tmp
bash tools/refresh_profiles5.sh > $O/refresh.log 2>&1
tail -5 $O/refresh.log | cut -c1-300
cd "$GRAFT_REPO_ROOT"
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
grep -n "passed\|failed" $O/tests.log | tail -3
( timeout 900 bash tools/run_asan.sh ) > $O/asan.log 2>&1
grep -v "^  File" $O/asan.log | grep -n "passed\|failed\|runtime error\|== \|asan:" | head -12 | cut -c1-220
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
