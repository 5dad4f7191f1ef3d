run() { echo "== $*"; env "$@" python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('one_batch_in_flight_images_per_sec'))"; }
run A=1
run YK_X_NS=2
run YK_X_CFG=0
run YK_X_CFG=0 YK_X_NS=2
run YK_X_DWLDS=20
run YK_X_DWLDS=64
run YK_X_SPLITK=1
run YK_X_NOSTEMFUSE=1
