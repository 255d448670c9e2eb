#!/bin/bash
# Sanitizer runs of the HOST side of libdeepsolid_hip.so (device code cannot be instrumented on gfx950 without xnack).
#   tools/asan_check.sh build     two instrumented builds (hipcc cross-compiles anywhere):
#                                 libdeepsolid_hip_asan.so  = AddressSanitizer + UBSan,  libdeepsolid_hip_ubsan.so = UBSan only
#   tools/asan_check.sh run-cpu   ASan+UBSan build under the no-GPU C-ABI tests (descriptor / workspace / Philox / error paths)
#   tools/asan_check.sh run       (GPU box, from the repo root) UBSan build under the LiH parity tests: create / workspace /
#                                 log psi / local energy / fused mcmc_step / gradient / destroy -> gpurun_out/ubsan_check.txt
# ASan cannot ride along on the GPU: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and aborts under the stock
# (uninstrumented) HIP runtime that torch ships.  The instrumented library is selected with DEEPSOLID_HIP_LIB.
set -u
cd "$(dirname "$0")/.."
RTDIR=$(dirname "$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)")
SAN="-O1 -g -Wno-option-ignored -shared-libsan -fno-omit-frame-pointer"
J=$(nproc)
case "${1:-run}" in
build)
    # (the library is twelve objects since round 4: the Makefile builds them with the sanitizer flags into their own directories)
    make -j$J -C deepsolid_amd/csrc OUT=../libdeepsolid_hip_asan.so OBJDIR=build_asan EXTRA="$SAN -fsanitize=address,undefined"
    make -j$J -C deepsolid_amd/csrc OUT=../libdeepsolid_hip_ubsan.so OBJDIR=build_ubsan EXTRA="$SAN -fsanitize=undefined -fno-sanitize-recover=undefined"
    ;;
run-cpu)
    export DEEPSOLID_HIP_LIB=$PWD/deepsolid_amd/libdeepsolid_hip_asan.so LD_PRELOAD=$RTDIR/libclang_rt.asan-x86_64.so
    export LD_LIBRARY_PATH=$RTDIR:${LD_LIBRARY_PATH:-} ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    python -m pytest tests/test_cabi_cpu.py -x -q -m "not gpu"
    ;;
run)
    mkdir -p gpurun_out
    export DEEPSOLID_HIP_LIB=$PWD/deepsolid_amd/libdeepsolid_hip_ubsan.so LD_LIBRARY_PATH=$RTDIR:${LD_LIBRARY_PATH:-}
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    python -m pytest tests/test_gpu_parity.py tests/test_gpu_vmc.py tests/test_gpu_grad.py -m gpu -x -q -k "lih and not lih_" \
        > gpurun_out/ubsan_check.txt 2>&1
    echo "exit code $?" >> gpurun_out/ubsan_check.txt
    echo "sanitizer reports: $(grep -c 'runtime error:' gpurun_out/ubsan_check.txt)" >> gpurun_out/ubsan_check.txt
    tail -5 gpurun_out/ubsan_check.txt
    ;;
esac
