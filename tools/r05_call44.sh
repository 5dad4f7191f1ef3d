#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c44; mkdir -p $O
( timeout 300 python tools/r05_h2d.py ) > $O/h2d.txt 2>&1; grep -v amdgpu.ids $O/h2d.txt
( timeout 600 python -m pytest tests/test_gpu_heads.py -q -x -m gpu ) > $O/tests.log 2>&1; tail -2 $O/tests.log
