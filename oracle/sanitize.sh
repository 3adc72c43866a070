#!/bin/bash
# AddressSanitizer + UBSan run of the CPU restatements and of the engine's arithmetic headers compiled for the host (oracle/Makefile,
# target `sanitize`).  TEST INFRASTRUCTURE: never part of the product, never run on the GPU box's card.
#   bash oracle/sanitize.sh [pytest args]        # default: tests/test_oracle.py
set -e
cd "$(dirname "$0")/.."
make -C oracle -s sanitize
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$rt" ] || rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export QLE_ORACLE_SAN=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ $# -eq 0 ]; then set -- tests/test_oracle.py; fi
LD_PRELOAD="$rt" python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider "$@"
