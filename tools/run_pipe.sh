python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py -x -q 2>&1 | grep -E "passed|failed"
python tools/kbench.py 32 2>&1 | sed -n 2,9p
NET=yolo_mobilev2 ALPHA=1.0 python tools/kbench.py 32 2>&1 | grep "B=32"
