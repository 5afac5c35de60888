#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the CPU test-suite (GPU sanitizers are not available on the pool): builds
# fluidgym_amd/csrc/_san/libfluidgym_hip_san.so (make san) and runs `pytest -m "not gpu"` with it.  What it covers: everything the
# library does on the host without a GPU -- argument checks, handles created with device < 0, the multi-block topology / coefficient
# table builders (fg_mb_topo.hip) on every mesh of tests/test_mb_tables.py, the sparse-operator and multilevel table entry checks.
#     bash tests/run_sanitizer_suite.sh [pytest args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/fluidgym_amd/csrc" san -j8 > /dev/null
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$R"
# the interpreter and torch are not instrumented: leak / ODR / allocator-pairing reports about them are switched off, everything else
# (heap / stack / global overflows, use after free, UB in the library) aborts the run
export ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0:detect_odr_violation=0:verify_asan_link_order=0:abort_on_error=1:protect_shadow_gap=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$ASAN" FLUIDGYM_AMD_LIB="$R/fluidgym_amd/csrc/_san/libfluidgym_hip_san.so" python -m pytest tests -m "not gpu" -q -x -p no:cacheprovider "$@"
