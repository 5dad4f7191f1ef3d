#!/bin/bash
# Host-side sanitizer runs of the C-ABI (need a GPU: nearly every entry point starts with a device query).
#   make -C k210_yolo_framework_amd/csrc asan ubsan && tools/run_asan.sh [pytest args]
# Two flavours, both through YK_LIB_PATH:
#   ubsan  libyolo_hip_ubsan.so carries its runtime: loads into the uninstrumented python as is (always run);
#   asan   libyolo_hip_asan.so needs the ASan runtime preloaded into python.  The HIP runtime of some images aborts during device
#          discovery with that runtime present (no report, `Fatal Python error: Aborted` inside torch.cuda.is_available); the script
#          probes that first and says so instead of failing the whole run.
# -s: a sanitizer report goes to stderr and the process aborts - pytest's capture would swallow it.
# Python itself is not instrumented: leak detection is off (the interpreter's own allocations would drown the report).
cd "$(dirname "$0")/.." || exit 1
TESTS=${@:-tests/test_gpu_graph.py tests/test_gpu_region.py tests/test_gpu_decode.py tests/test_gpu_persist.py tests/test_abi.py -m gpu -q -x -s}
CS=$PWD/k210_yolo_framework_amd/csrc
rc=0
echo "== ubsan"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 YK_LIB_PATH=$CS/libyolo_hip_ubsan.so python -m pytest $TESTS || rc=1
echo "== asan"
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:detect_odr_violation=0
if LD_PRELOAD=$RT python -c "import torch, sys; sys.exit(0 if torch.cuda.is_available() else 3)" > /tmp/asan_probe.log 2>&1; then
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 YK_LIB_PATH=$CS/libyolo_hip_asan.so LD_PRELOAD=$RT python -m pytest $TESTS || rc=1
else
    echo "asan: the HIP runtime does not come up with the ASan runtime preloaded on this box (probe: $(tail -1 /tmp/asan_probe.log | cut -c1-120)); skipped"
    for v in "HSA_XNACK=1" "HSA_ENABLE_SDMA=0" "ASAN_OPTIONS=$ASAN_OPTIONS:handle_abort=0:handle_segv=0:use_sigaltstack=0"; do
        if env $v LD_PRELOAD=$RT python -c "import torch, sys; sys.exit(0 if torch.cuda.is_available() else 3)" > /tmp/asan_probe2.log 2>&1; then
            echo "asan: comes up with $v"
            env $v UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 YK_LIB_PATH=$CS/libyolo_hip_asan.so LD_PRELOAD=$RT python -m pytest $TESTS || rc=1
            break
        else
            echo "asan: not with $v either: $(grep -v '^  File' /tmp/asan_probe2.log | head -3 | cut -c1-160 | tr '\n' '|')"
        fi
    done
fi
exit $rc
