#!/bin/bash
# round 5, call 46: the ring GEMM with the weight fragments in registers (yk_igemm_br.h): parity of the f16 mode with it on, Darknet-53 per layer
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c46; mkdir -p $O
( YK_PIPE_BR=1 timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py -q -x -m gpu ) > $O/tests_br.log 2>&1; tail -2 $O/tests_br.log
for v in 0 1; do
( YK_PIPE_BR=$v timeout 300 python tools/darknet_layers.py f16 32 ) > $O/darknet_f16_32_br$v.txt 2>&1; grep "launches" $O/darknet_f16_32_br$v.txt
done
grep "conv3x3s1_128to256\|conv3x3s1_256to512\|conv3x3s1_512to1024\|conv1x1s1_256to128" $O/darknet_f16_32_br0.txt | sort -u -k6 | head -8
echo ---
grep "conv3x3s1_128to256\|conv3x3s1_256to512\|conv3x3s1_512to1024\|conv1x1s1_256to128" $O/darknet_f16_32_br1.txt | sort -u -k6 | head -8
