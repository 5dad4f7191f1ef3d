#!/bin/bash
# GPU call 1 of round 5: diagnostics only (tools/r05_diag1.py, Darknet at B=32, the new graph tests)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c1; mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_e2e.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout 300 python tools/r05_diag1.py a b c ) > $O/diag1.txt 2>&1; tail -40 $O/diag1.txt
( GPU_MAX_HW_QUEUES=16 timeout 120 python tools/r05_diag1.py b ) > $O/diag1_q16.txt 2>&1; tail -18 $O/diag1_q16.txt
( timeout 200 python tools/darknet_layers.py f16 32 ) > $O/darknet_f16_b32.txt 2>&1; head -12 $O/darknet_f16_b32.txt
( timeout 200 python tools/darknet_layers.py f16x2 32 ) > $O/darknet_x2_b32.txt 2>&1; head -6 $O/darknet_x2_b32.txt
