python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | grep -E "passed|failed"
for sh in "26 26 256 512" "52 52 128 256" "13 13 512 1024" "104 104 64 128"; do
    echo -n "shape $sh: "; PROFILE=1 python tools/igemm_one.py $sh 16 2>&1 | grep "to512\[\|to1024\[\|to256\[\|to128\[" | tail -1
done
python tools/kbench.py 32 2>&1 | grep "768to192\|reduce.*192\|512to128\|reduce.*128+\|768to128\|B=32"
NET=yolo SHAPE=416,416 ALPHA=1.0 python tools/kbench.py 16 2>&1 | grep "B=16"
NET=tiny_yolo SHAPE=416,416 ALPHA=1.0 python tools/kbench.py 64 2>&1 | grep "B=64"
