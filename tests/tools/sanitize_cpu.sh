#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (the C host library and the oracle), through the CPU test suite.
# GPU sanitizers are not available on this pool; the device code is covered by the parity tests instead.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
asan=$(gcc -print-file-name=libasan.so)
san="-O1 -g -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared"
gcc -std=gnu99 $san -I$root/include -I$root/cpecan-signal_amd/csrc -I$root/cpecan-signal_amd/csrc/host \
    -o /tmp/libcpecan_host_asan.so $root/cpecan-signal_amd/csrc/host/cpecan_api.c \
    $root/cpecan-signal_amd/csrc/host/cpecan_internals.c -L$root/cpecan-signal_amd -lcpecan_hip \
    -Wl,-rpath,$root/cpecan-signal_amd -lm -lpthread
# the C caller of the reference-shaped API (tests/c/reference_api_test.c), host-only part, against the sanitized library
gcc -std=gnu99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include \
    $root/tests/c/reference_api_test.c -o /tmp/reference_api_test_asan /tmp/libcpecan_host_asan.so \
    -L$root/cpecan-signal_amd -lcpecan_hip -Wl,-rpath,$root/cpecan-signal_amd -Wl,-rpath,/tmp -lm -lpthread
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 /tmp/reference_api_test_asan cpu $root/tests/golden \
    > /tmp/sanitize_c_program.log 2>&1 || { tail -20 /tmp/sanitize_c_program.log; exit 1; }
grep -c "^ok " /tmp/sanitize_c_program.log
gcc -std=gnu99 -fno-fast-math $san -o /tmp/liborc_asan.so $root/oracle/cpecan_oracle.c -lm
cd $root
CPECAN_HOST_LIB=/tmp/libcpecan_host_asan.so CPECAN_ORACLE_LIB=/tmp/liborc_asan.so LD_PRELOAD=$asan \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/sanitize_cpu.log | tail -3
n=$(cat /tmp/sanitize_cpu.log /tmp/sanitize_c_program.log | grep -c "AddressSanitizer\|runtime error" || true)
echo "sanitizer reports: $n"
[ "$n" = 0 ]
