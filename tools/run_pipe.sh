python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | tail -3
python tools/kbench.py 32 2>&1 | tail -24
python tools/phase.py 9 32 2>&1 | tail -11
python tools/phase.py 6 32 2>&1 | tail -11
