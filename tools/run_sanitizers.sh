#!/bin/bash
# CPU sanitizer runs of the native host code.  Never on the GPU box: the GPU pool refuses sanitizer runs,
# and nothing here touches a GPU.  usage: bash tools/run_sanitizers.sh [summary file, default profiles/r04_sanitizers.txt]
#   1. liblrf_pack.so (lrf_amd/csrc/lrf_pack.cpp: threaded C++ that parses untrusted streams) with -fsanitize=address,undefined
#      and, separately, -fsanitize=thread (the worker pool + fork handler);
#   2. the oracle (oracle/lrf_oracle.c + lrf_oracle_any.c) with -fsanitize=address,undefined (make -C oracle san);
#   3. the CPU tests that drive them — tests/test_container_abi.py, the native / repack / unpack tests of tests/test_anyshape.py,
#      tests/test_oracle_golden.py — under LD_PRELOAD of the sanitizer runtime (python itself is not instrumented).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/profiles/r04_sanitizers.txt}
cd "$REPO"
SRC=lrf_amd/csrc/lrf_pack.cpp
ASAN_LIB=lrf_amd/liblrf_pack_asan.so
TSAN_LIB=lrf_amd/liblrf_pack_tsan.so
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $ASAN_LIB $SRC -lz || exit 1
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fno-omit-frame-pointer -fsanitize=thread -o $TSAN_LIB $SRC -lz || exit 1
make -C oracle san -s || exit 1
LIBASAN=$(gcc -print-file-name=libasan.so)
LIBUBSAN=$(gcc -print-file-name=libubsan.so)
LIBTSAN=$(gcc -print-file-name=libtsan.so)
{
  echo "# CPU sanitizer runs, $(date -u +%Y-%m-%dT%H:%MZ), gcc $(gcc -dumpversion), $(python3 -c 'import sys; print("python", sys.version.split()[0])')"
  echo "## 1. liblrf_pack.so with AddressSanitizer + UBSan: tests/test_container_abi.py + the packer tests of tests/test_anyshape.py"
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$LIBASAN $LIBUBSAN" LRF_PACK_LIB=$REPO/$ASAN_LIB \
    python3 -m pytest tests/test_container_abi.py tests/test_anyshape.py -q -m "not gpu" -k "native or repack or unpack or pack or fork or crafted or stream or abi or container" -p no:cacheprovider 2>&1 | tail -4
  echo "## 2. liblrf_pack.so with ThreadSanitizer: the same tests without the fork test (TSan cannot follow a fork of a"
  echo "##    multi-threaded process: that test hangs under it and runs under ASan above); OPENBLAS_NUM_THREADS=1 keeps numpy's own"
  echo "##    (uninstrumented) BLAS threads out of the report"
  OPENBLAS_NUM_THREADS=1 TSAN_OPTIONS=report_signal_unsafe=0:halt_on_error=0:die_after_fork=0 LD_PRELOAD="$LIBTSAN" LRF_PACK_LIB=$REPO/$TSAN_LIB \
    timeout 900 python3 -m pytest tests/test_container_abi.py tests/test_anyshape.py -q -m "not gpu" --timeout 300 \
    -k "(native or repack or unpack or pack or crafted or stream or abi or container) and not fork" -p no:cacheprovider 2>&1 | tail -6
  echo "## 3. the oracle with AddressSanitizer + UBSan: tests/test_oracle_golden.py tests/test_qmf_kwargs.py tests/test_identity_rate.py"
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$LIBASAN $LIBUBSAN" LRF_ORACLE_SO=$REPO/oracle/_build/liblrf_oracle_san.so \
    timeout 1500 python3 -m pytest tests/test_oracle_golden.py tests/test_qmf_kwargs.py tests/test_identity_rate.py -q -m "not gpu" --timeout 600 -p no:cacheprovider 2>&1 | tail -4
} | tee "$OUT"
