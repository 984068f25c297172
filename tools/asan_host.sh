#!/bin/bash
# Memory and undefined-behaviour check of the library's HOST side: the C-ABI driver and the library's own sources in
# one executable, host code built with AddressSanitizer + UBSan (device code is not instrumented).
#   here:            bash tools/asan_host.sh build
#   on the GPU box:  bash tools/asan_host.sh run      (expects the binary built here; prints the sanitizer report, if any)
set -e
cd "$(dirname "$0")/.."
case "$1" in
build)
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
        -Iinclude radiative-transfer_amd/csrc/rt_api.hip radiative-transfer_amd/csrc/rt_kernels.hip \
        radiative-transfer_amd/csrc/rt_shell.hip tests/native/abi_driver.cpp -o tests/native/abi_driver_asan
    ;;
run)
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1 timeout -k 10 240 tests/native/abi_driver_asan
    ;;
*)
    echo "usage: $0 build|run"; exit 2;;
esac
