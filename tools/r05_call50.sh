#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c50; mkdir -p $O
( timeout 1500 python -m pytest tests -q -x -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
