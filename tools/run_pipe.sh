python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | grep -E "passed|failed"
python tools/kbench.py 32 2>&1 | sed -n 2,24p
