#!/bin/bash
# sanitize_cpu.sh - the host code under sanitizers, on the CPU (GPU sanitizers are not available on this pool):
#   asan   AddressSanitizer + UndefinedBehaviorSanitizer builds of the library (host side) and of all the tools; the CPU tests
#          of the host-only entry points through Python (packing, vsc_windows_build, vsc_sam_order, ABI) and of the tools
#          (fasta_writer, vcf_loader, bidir_index, the CLI error paths of the others)
#   tsan   ThreadSanitizer build: vsc_windows_build (VCF parsed in chunks on all threads, blocks assembled independently, bit
#          streams stitched at shared boundary words with atomics) and the tools' threaded packing (bidir_index)
# Builds go to build/sanitize/ (not tracked); the log to profiles/<TAG>_sanitizers_cpu.txt.
# vsc_multi.cpp's engine needs devices: its threads run under the sanitizers over a host stand-in, tools/multi_tsan/run.sh.
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=${TAG:-r04}; LOG=$ROOT/profiles/${TAG}_sanitizers_cpu.txt
HIPCC=/opt/rocm/bin/hipcc; CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")
SRCS="vsc_kernels.hip vsc_seed.hip vsc_sort.hip vsc_api.cpp vsc_pack.cpp vsc_windows.cpp vsc_multi.cpp"
TOOLS="bidir_index bidir_mapping vcf_loader bam_merger_ref_only bam_merger fasta_writer classification_pipeline varscot_pipeline"
: > "$LOG"
build() {  # build NAME "sanitizer flags"
    local name=$1 flags=$2 out=$ROOT/build/sanitize/$1
    mkdir -p "$out/bin" "$out/obj"
    ( cd "$ROOT/varscot_amd/csrc"
      for s in $SRCS; do
          $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -I"$ROOT/include" -I. $flags -fno-gpu-sanitize -shared-libsan \
              -x hip -c "$s" -o "$out/obj/${s%.*}.o" &
      done; wait
      $HIPCC --offload-arch=gfx950 $flags -fno-gpu-sanitize -shared-libsan "$out"/obj/*.o -pthread -ldl -shared -o "$out/libvarscot_hip.so" || exit 1
      for t in $TOOLS; do
          $CXX -O1 -g -std=c++17 $flags -shared-libsan "tools/$t.cpp" -I"$ROOT/include" -L"$out" -lvarscot_hip -pthread \
              -Wl,-rpath,"$out" -Wl,-rpath,"$RT" -Wl,-rpath-link,/opt/rocm/lib -o "$out/bin/$t" &
      done; wait ) 2>&1 | grep -E "error|Error" | head -5
    ls "$out/bin" | wc -l
}
run() {  # run NAME runtime.so "options env" tests...
    local name=$1 rt=$2 opts=$3; shift 3
    echo "== $name: $*" | tee -a "$LOG"
    ( cd "$ROOT" && env $opts VSC_LIB_PATH=$ROOT/build/sanitize/$name/libvarscot_hip.so VSC_TEST_BIN=$ROOT/build/sanitize/$name/bin \
        VSC_NO_TORCH_PRELOAD=1 LD_PRELOAD=$RT/$rt OMP_NUM_THREADS=4 timeout 1500 python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider "$@" ) 2>&1 \
        | grep -vE "^\s*$" | tail -25 | tee -a "$LOG"
}
case ${1:-all} in
asan|all)
    echo "# AddressSanitizer + UndefinedBehaviorSanitizer (host code; clang $($CXX --version | head -1))" | tee -a "$LOG"
    build asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined"
    run asan libclang_rt.asan-x86_64.so "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1" \
        tests/test_abi.py tests/test_variants.py tests/test_tools.py ;;&
tsan|all)
    echo "# ThreadSanitizer (host code)" | tee -a "$LOG"
    build tsan "-fsanitize=thread"
    run tsan libclang_rt.tsan-x86_64.so "TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0" tests/test_variants.py tests/test_tools.py ;;
esac
grep -cE "ERROR: (Address|Thread)Sanitizer|runtime error:|WARNING: ThreadSanitizer" "$LOG" | sed 's/^/sanitizer reports: /' | tee -a "$LOG"
rm -rf "$ROOT/build/sanitize"  # (50 MB that would travel to the GPU box with every gpurun call)
