#!/bin/bash
# run.sh - host code that needs a device, under the sanitizers on the CPU (GPU sanitizers are not available on this pool):
#   1. csrc/vsc_multi.cpp (the multi-device engine as it is shipped) linked with a host stand-in of the device layer;
#   2. the mergers (bam_merger_ref_only, bam_merger) linked with invented scores instead of the library's scoring calls;
#   3. varscot_pipeline with the real vsc_windows.cpp / vsc_pack.cpp over a brute-force host search and invented scores.
# ThreadSanitizer, then AddressSanitizer + UBSan.  Builds in a temporary directory; the log goes to
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
# ---- the mergers: their own host code (parsing, filters, coordinate restoration, threaded text) over invented scores ----------
python3 "$ROOT/tests/make_merger_inputs.py" "$OUT/sc" 20240 > /dev/null || { echo "could not make the mergers' inputs" | tee -a "$LOG"; exit 1; }
S=$OUT/sc
for kind in address,undefined thread; do
    echo "# -fsanitize=$kind: bam_merger_ref_only / bam_merger over tools/multi_tsan/stub_scores.cpp + csrc/vsc_pack.cpp (FASTA text and packed genomes, MIT / feature matrix / forest)" | tee -a "$LOG"
    for t in bam_merger_ref_only bam_merger; do
        $CXX $FLAGS -fsanitize=$kind -fno-sanitize-recover=undefined "$ROOT/varscot_amd/csrc/tools/$t.cpp" "$HERE/stub_scores.cpp" "$ROOT/varscot_amd/csrc/vsc_pack.cpp" \
            -pthread -o "$OUT/$t" 2>&1 | grep -E "error" | head -5 | tee -a "$LOG"
    done
    for packed in "" "VARSCOT_PACKED_GENOME=$S/genome_idx VARSCOT_PACKED_SNP_GENOME=$S/snp_idx"; do
        for mit in 0 1; do
            env $packed TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 VARSCOT_TRACE=1 "$OUT/bam_merger_ref_only" "$S/o1.txt" "$S/f1.txt" "$S/ref.sam" \
                "$S/targets.bed" "$S/genome.fa" "$S/activity.txt" 5 23 $mit 2>&1 | grep -vE "^(Read|Filter|Write|Process|Merge|Compute|Calculate)" | tail -20 | tee -a "$LOG"
            env $packed TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 VARSCOT_TRACE=1 "$OUT/bam_merger" "$S/o2.txt" "$S/f2.txt" "$S/ref.sam" "$S/snp.sam" \
                "$S/targets.bed" "$S/genome.fa" "$S/snp.fa" "$S/activity.txt" 5 23 2 $mit 2>&1 | grep -vE "^(Read|Filter|Write|Process|Merge|Compute|Calculate)" | tail -20 | tee -a "$LOG"
            echo "rows: $(($(wc -l < "$S/o1.txt") - 1)) reference-only, $(($(wc -l < "$S/o2.txt") - 1)) merged (mit $mit${packed:+, packed genomes})" | tee -a "$LOG"
        done
    done
done
# ---- varscot_pipeline (the driver's stages in one process): its own code + the real vsc_windows.cpp / vsc_pack.cpp over a brute-force
# host search and invented scores (stub_search.cpp, stub_scores.cpp) -------------------------------------------------------------------
for kind in address,undefined thread; do
    echo "# -fsanitize=$kind: varscot_pipeline + csrc/vsc_windows.cpp + csrc/vsc_pack.cpp over tools/multi_tsan/{stub_search,stub_scores}.cpp" | tee -a "$LOG"
    $CXX $FLAGS -fsanitize=$kind -fno-sanitize-recover=undefined "$ROOT/varscot_amd/csrc/tools/varscot_pipeline.cpp" "$HERE/stub_scores.cpp" "$HERE/stub_search.cpp" \
        "$ROOT/varscot_amd/csrc/vsc_pack.cpp" "$ROOT/varscot_amd/csrc/vsc_windows.cpp" -pthread -o "$OUT/varscot_pipeline" 2>&1 | grep -E "error" | head -5 | tee -a "$LOG"
    for args in "-e mit" "-e prob" "-e class -S 1" "-e mit -f $S/in.vcf -s 0 -t 3" "-e prob -f $S/in.vcf -s 0,0 -t 2 -p AG"; do
        rm -f "$S"/vp*.txt
        env TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 VARSCOT_RF_MODEL="$ROOT/varscot_amd/models/rfClassifier.vscrf" \
            "$OUT/varscot_pipeline" -b "$S/targets.bed" -g "$S/genome.fa" -i "$S/genome_idx" -o "$S/vp" -a "$S/activity.txt" -m 5 $args 2>&1 | tail -20 | tee -a "$LOG"
        echo "varscot_pipeline ${args//$S\//}: $(cat "$S"/vp*.txt 2>/dev/null | grep -vc '^#') rows" | tee -a "$LOG"
    done
    # bidir_mapping (reads, SAM order, MD strings, threaded SAM text) over the same stand-in search; classification_pipeline
    # (feature matrix parsed back, Score column rewritten) over the invented votes
    for t in bidir_mapping classification_pipeline; do
        $CXX $FLAGS -fsanitize=$kind -fno-sanitize-recover=undefined "$ROOT/varscot_amd/csrc/tools/$t.cpp" "$HERE/stub_scores.cpp" "$HERE/stub_search.cpp" \
            "$ROOT/varscot_amd/csrc/vsc_pack.cpp" "$ROOT/varscot_amd/csrc/vsc_windows.cpp" -pthread -o "$OUT/$t" 2>&1 | grep -E "error" | head -5 | tee -a "$LOG"
    done
    for style in 0 1; do
        env TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/bidir_mapping" -G "$S/genome.fa" -I "$S/genome_idx" -R "$S/targets.fa" -M 5 -T 4 \
            -O "$S/bm.sam" --md-style $style 2>&1 | grep -vE "^(Reads loaded|Index loaded)" | tail -20 | tee -a "$LOG"
        echo "bidir_mapping --md-style $style: $(wc -l < "$S/bm.sam") SAM records" | tee -a "$LOG"
    done
    "$OUT/bam_merger_ref_only" "$S/cp.txt" "$S/cp_feature_matrix.txt" "$S/ref.sam" "$S/targets.bed" "$S/genome.fa" "$S/activity.txt" 5 23 1 > /dev/null 2>&1
    env TSAN_OPTIONS=halt_on_error=0 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 VARSCOT_RF_MODEL="$ROOT/varscot_amd/models/rfClassifier.vscrf" \
        "$OUT/classification_pipeline" "$S/cp.txt" "$S/cp_feature_matrix.txt" FALSE 2>&1 | tail -20 | tee -a "$LOG"
    echo "classification_pipeline: $(grep -vc '^#' "$S/cp.txt") rows rewritten" | tee -a "$LOG"
done
grep -cE "ERROR: (Address|Leak)Sanitizer|runtime error:|WARNING: ThreadSanitizer|FAIL " "$LOG" | sed 's/^/sanitizer reports + failures: /' | tee -a "$LOG"
cat "$HERE/NOTES.txt" >> "$LOG"
