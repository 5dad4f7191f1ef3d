#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c18; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py tests/test_gpu_e2e.py -q -m gpu ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
for lc in 0 1; do for b in 32 64; do
( YK_IGEMM_LC=$lc timeout 200 python tools/darknet_layers.py f16 $b ) 2>&1 | grep -v amdgpu | head -2 | tail -1 | sed "s/^/LC=$lc /"
done; done
( YK_IGEMM_LC=1 timeout 200 python tools/darknet_layers.py f16 32 ) > $O/darknet_f16_b32.txt 2>&1
( YK_IGEMM_LC=1 timeout 200 python tools/darknet_layers.py f16 64 ) > $O/darknet_f16_b64.txt 2>&1
( timeout 200 python tools/netbench.py tiny_yolo yolo ) 2>&1 | grep "^|"
