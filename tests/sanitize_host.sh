#!/bin/bash
# CPU only: the C host side (hercules_amd/csrc/hq_host.c) under AddressSanitizer + UBSan, driven by
# the host-logic tests.  (GPU sanitizers are not available on this pool; the HIP library is used
# as built.)   bash tests/sanitize_host.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
python -m hercules_amd.build > /dev/null
OUT=${TMPDIR:-/tmp}/libhq_host_asan.so
gcc -O1 -g -std=gnu99 -fPIC -shared -fvisibility=hidden -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I include -o "$OUT" hercules_amd/csrc/hq_host.c -L hercules_amd/csrc -lhq_solver \
    -Wl,-rpath,"$PWD/hercules_amd/csrc" -lm
export HQ_HOST_LIB="$OUT"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
exec python -m pytest tests/test_host_partition.py -q -x -p no:cacheprovider "$@"
