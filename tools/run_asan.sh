#!/bin/bash
# Host-side sanitizer run of the C-ABI (needs a GPU: nearly every entry point starts with a device query).
#   make -C k210_yolo_framework_amd/csrc asan && tools/run_asan.sh [pytest args]
# Python itself is not instrumented: leak detection is off (the interpreter's own allocations would drown the report) and the ASan
# runtime is preloaded so that the instrumented library finds it.
cd "$(dirname "$0")/.." || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export YK_LIB_PATH=$PWD/k210_yolo_framework_amd/csrc/libyolo_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT python -m pytest ${@:-tests/test_gpu_graph.py tests/test_gpu_region.py tests/test_gpu_decode.py tests/test_gpu_persist.py tests/test_abi.py -m gpu -q -x}
