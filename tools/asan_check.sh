#!/bin/bash
# Sanitizer run of the HOST side of libdeepsolid_hip.so (AddressSanitizer + UBSan; device code cannot be instrumented on gfx950).
#   build (anywhere, hipcc cross-compiles):   tools/asan_check.sh build
#   run (on the GPU box, from the repo root): tools/asan_check.sh run        -> gpurun_out/asan_check.txt
# The run drives the C ABI through the ordinary parity tests of a small system (create / workspace / log psi / local energy /
# fused mcmc_step / gradient / destroy) with the instrumented library selected by DEEPSOLID_HIP_LIB.
set -u
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LIB=$PWD/deepsolid_amd/libdeepsolid_hip_asan.so
case "${1:-run}" in
build)
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-result -Wno-option-ignored \
        -fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer deepsolid_amd/csrc/ds_api.hip -o "$LIB"
    ;;
run)
    mkdir -p gpurun_out
    export DEEPSOLID_HIP_LIB=$LIB LD_PRELOAD=$RT LD_LIBRARY_PATH=$(dirname "$RT"):${LD_LIBRARY_PATH:-}
    # (python itself and the HIP runtime are not leak-clean: leak detection off; every other report is fatal)
    export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    python -m pytest tests/test_gpu_parity.py tests/test_gpu_vmc.py tests/test_gpu_grad.py -m gpu -x -q -k "lih and not lih_" \
        > gpurun_out/asan_check.txt 2>&1
    echo "exit code $?" >> gpurun_out/asan_check.txt
    grep -c "ERROR: AddressSanitizer\|runtime error:" gpurun_out/asan_check.txt | sed 's/^/sanitizer reports: /' >> gpurun_out/asan_check.txt
    tail -5 gpurun_out/asan_check.txt
    ;;
esac
