#!/bin/bash
# round-4 GPU call 1: graph pipeline tests + bench line + tile / fusion sweeps of the developer build (planning data for the kernel work)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
O=gpurun_out/c1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_abi.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -25 ) > $O/tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $O/smoke.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err )
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so
( timeout 900 python tools/xsweep.py "YK_X_CFG=3" "YK_X_CFG=3 YK_X_NS=3" "YK_X_CFG=1 YK_X_NS=4" "YK_X_NS=4" "YK_X_SPLITK=1" "YK_X_SPLITK=2" \
      "YK_XB_ALWAYS=1" "YK_XB_ALWAYS=1 YK_XB_DB=1" "YK_XB_ALWAYS=1 YK_XB_TM=2" "YK_XB_ALWAYS=1 YK_XB_TM=2 YK_XB_DB=1" ) > $O/xsweep.log 2>&1
tail -30 $O/tests.log; cat $O/smoke.log; cut -c1-1500 $O/bench.json; tail -5 $O/bench.err
