#!/bin/bash
# round-4 GPU call 3: operand-path probe, the gpu tests touched so far, the mAP test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
O=gpurun_out/c3
export TMPDIR=/tmp
( timeout 300 tools/dma_probe.bin ) > $O/dma_probe.log 2>&1
( timeout 1500 python -m pytest tests/test_abi.py tests/test_gpu_graph.py tests/test_gpu_map.py tests/test_gpu_e2e.py -m gpu -q 2>&1 | tail -40 ) > $O/tests.log 2>&1
cat $O/dma_probe.log; tail -30 $O/tests.log; cat gpurun_out/map_eval.json
