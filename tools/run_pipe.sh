python -m pytest tests/test_gpu_layers.py tests/test_gpu_net.py -x -q 2>&1 | tail -2
python tools/kbench.py 32 2>&1 | tail -24
NET=yolo SHAPE=416,416 ALPHA=1.0 python tools/kbench.py 16 2>&1 | grep "B=16"
NET=tiny_yolo SHAPE=416,416 ALPHA=1.0 python tools/kbench.py 64 2>&1 | grep "B=64"
