#!/bin/bash
# run.sh - csrc/vsc_multi.cpp (the multi-device engine as it is shipped) linked with a host stand-in of the device layer and
# run under ThreadSanitizer, then under AddressSanitizer + UBSan, on the CPU.  Builds in a temporary directory; the log goes to
# profiles/<TAG>_multi_tsan.txt.   usage: tools/multi_tsan/run.sh
set -o pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); TAG=${TAG:-r04}; LOG=$ROOT/profiles/${TAG}_multi_tsan.txt
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT=$(mktemp -d); trap 'rm -rf "$OUT"' EXIT
FLAGS="-std=c++17 -O1 -g -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/varscot_amd/csrc -I$HERE"
: > "$LOG"
for kind in thread address,undefined; do
    echo "# -fsanitize=$kind: vsc_multi.cpp + tools/multi_tsan/{stub_device,driver}.cpp ($($CXX --version | head -1))" | tee -a "$LOG"
    $CXX $FLAGS -fsanitize=$kind -fno-sanitize-recover=undefined "$ROOT/varscot_amd/csrc/vsc_multi.cpp" "$HERE/stub_device.cpp" "$HERE/driver.cpp" \
        -pthread -ldl -o "$OUT/multi_$$" 2>&1 | grep -E "error" | head -5 | tee -a "$LOG"
    for rep in 1 2 3; do
        TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 timeout 600 "$OUT/multi_$$" 2>&1 | tail -40 | tee -a "$LOG"
    done
done
grep -cE "ERROR: (Address|Leak)Sanitizer|runtime error:|WARNING: ThreadSanitizer|FAIL " "$LOG" | sed 's/^/sanitizer reports + failures: /' | tee -a "$LOG"
