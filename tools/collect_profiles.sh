#!/bin/bash
# collect_profiles.sh TAG - run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh r01'):
# bench lines, rocprofv3 kernel statistics and PMC counters of the search kernel for c2 and c3.
# Counters are collected in their own passes (no tracing alongside), one group per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Results land in gpurun_out/$TAG/ and are digested
# by tools/digest_profiles.py into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in c2 c3; do
    ARGS="$ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline"
    timeout -k 10 400 python3 $ARGS > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
    echo "bench $W done"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$W" -o run --output-format csv -- python3 $ARGS > "$OUT/stats_$W.json" 2> "$OUT/stats_$W.err"
    echo "stats $W done"
    timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        -d "$OUT/pmc_sq_$W" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_sq_$W.err"
    echo "pmc sq $W done"
    timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$W" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_fetch_$W.err"
    timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_$W" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_write_$W.err"
    echo "pmc hbm $W done"
done
# the per-dispatch counter tables are large: keep only the rows of our kernels
for f in "$OUT"/pmc_*/*counter_collection.csv "$OUT"/pmc_*/*/*counter_collection.csv; do
    [ -f "$f" ] || continue
    { head -1 "$f"; grep -E 'vsc::' "$f" || true; } > "$f.small" && mv "$f.small" "$f"
done
# bench lines of the other workloads (no profiling)
timeout -k 10 500 python3 $ROOT/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --classify > "$OUT/bench_c5_classify.json" 2> "$OUT/bench_c5_classify.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c3 --steps 3 --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || true
find "$OUT" -name '*kernel_trace.csv' -delete
du -sh "$OUT"
