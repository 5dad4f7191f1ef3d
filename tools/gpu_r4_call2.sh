#!/bin/bash
# round-4 GPU call 2: whole-line pixel operand of xg_kernel - parity, then per-launch A/B against the round-3 form (developer builds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
O=gpurun_out/c2
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_graph.py tests/test_gpu_net.py tests/test_gpu_e2e.py -m gpu -q 2>&1 | tail -40 ) > $O/tests.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( timeout 300 python tools/xbench.py ) > $O/xbench_new.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev_base.so
( timeout 300 python tools/xbench.py ) > $O/xbench_base.log 2>&1
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( timeout 600 python tools/xsweep.py "YK_X_SPLITK=1" "YK_X_CFG=3 YK_X_NS=3" "YK_X_NS=3" ) > $O/xsweep.log 2>&1
unset YK_LIB_PATH
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err )
tail -15 $O/tests.log; paste <(cut -c1-75 $O/xbench_base.log) <(cut -c60-75 $O/xbench_new.log); cut -c1-700 $O/bench.json
