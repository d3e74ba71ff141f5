#!/bin/bash
# AddressSanitizer on the HOST side of the C-ABI library (sanitizers; GPU ASan is not available on the pool).
# Builds og_api.hip with -fsanitize=address for the host code only (-fno-gpu-sanitize: the gfx950 code object is the product's)
# into build/ (git-ignored, never loaded by default) and runs the CPU tests that exercise the library without a GPU -- handle
# life cycle, state_dict intake (copies, shape / key / dtype checks, call order), option parsing, every export -- under it.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
mkdir -p build
"$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -fno-gpu-sanitize -shared-libasan -Wno-unused-function \
    -o build/libopenglottal_hip_asan.so openglottal_amd/csrc/og_api.hip
OPENGLOTTAL_HIP_LIB="$PWD/build/libopenglottal_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 \
    python3 -m pytest tests/test_abi_errors.py tests/test_host_logic.py tests/test_launch_plan.py -q -p no:cacheprovider
