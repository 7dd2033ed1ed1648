#!/bin/bash
# collect_profiles.sh TAG - run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh r01'):
# bench lines, rocprofv3 kernel statistics and PMC counters of the search kernel for c2 and c3.
# Counters are collected in their own passes (no tracing alongside), one group per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Results land in gpurun_out/$TAG/ and are digested
# by tools/digest_profiles.py into profiles/.
set -e -o pipefail
TAG=${1:-r01}
PART=${2:-all}   # pmc | c5 | forest | lines | all - gpurun gives a call at most 20 minutes: one part per call
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$PART" == pmc ] || [ "$PART" == all ]; then
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
fi
if [ "$PART" == c5 ] || [ "$PART" == all ]; then
# c5 (two batches of 10 000 reads are enough for the counters): the kernels that write the per-hit rows, both routes
for R in fused two-pass; do
    ARGS="$ROOT/bench.py --workload c5 --guides 20000 --steps 1 --warmup 1 --rows $R --no-cpu-baseline"
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 400 rocprofv3 --pmc $C -d "$OUT/pmc_c5_${R}_$C" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_c5_${R}_$C.err" || true
    done
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c5_$R" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/stats_c5_$R.err" || true
    echo "pmc c5 $R done"
done
for f in "$OUT"/pmc_c5_*/*counter_collection.csv "$OUT"/pmc_c5_*/*/*counter_collection.csv; do
    [ -f "$f" ] || continue
    { head -1 "$f"; grep -E 'vsc::' "$f" || true; } > "$f.small" && mv "$f.small" "$f"
done
fi
if [ "$PART" == forest ] || [ "$PART" == all ]; then
# the forest walk (c5 with the classifier, one batch of 10 000 reads + the warm-up batch): vector and LDS instruction counts,
# LDS cycles and bank-conflict cycles of rf_predict_kernel, pair nodes and (hook rf_form = 1) the compact nodes of round 3
for F in pair compact; do
    HOOK=""; [ "$F" == compact ] && HOOK="--hook rf_form=1"
    ARGS="$ROOT/bench.py --workload c5 --classify --guides 10000 --steps 1 --warmup 1 --no-cpu-baseline $HOOK"
    timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/pmc_forest_${F}_a" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_forest_${F}_a.err" || true
    timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/pmc_forest_${F}_b" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_forest_${F}_b.err" || true
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_forest_$F" -o run --output-format csv -- python3 $ARGS > "$OUT/bench_forest_$F.json" 2> "$OUT/stats_forest_$F.err" || true
    echo "forest $F done"
done
for f in "$OUT"/pmc_forest_*/*counter_collection.csv "$OUT"/pmc_forest_*/*/*counter_collection.csv; do
    [ -f "$f" ] || continue
    { head -1 "$f"; grep -E 'vsc::' "$f" || true; } > "$f.small" && mv "$f.small" "$f"
done
fi
if [ "$PART" == lines ] || [ "$PART" == all ]; then
# bench lines of the other workloads (no profiling)
timeout -k 10 500 python3 $ROOT/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --rows two-pass --no-cpu-baseline > "$OUT/bench_c5_two_pass.json" 2> "$OUT/bench_c5_two_pass.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --classify > "$OUT/bench_c5_classify.json" 2> "$OUT/bench_c5_classify.err" || true
# the product's own multi-device driver (one process over the devices behind the C ABI) on this one GPU: one context, and a
# rehearsal with four contexts on the device (device copies in place of RCCL)
timeout -k 10 400 python3 $ROOT/bench.py --multi abi --gpus 1 --workload c3 --steps 5 --warmup 2 > "$OUT/bench_abi_c3_x1.json" 2> "$OUT/bench_abi_c3_x1.err" || true
timeout -k 10 400 python3 $ROOT/bench.py --multi abi --gpus 4 --abi-devices 0,0,0,0 --workload c3 --steps 5 --warmup 2 > "$OUT/bench_abi_c3_x4_one_gpu.json" 2> "$OUT/bench_abi_c3_x4_one_gpu.err" || true
timeout -k 10 400 python3 $ROOT/bench.py --multi abi --gpus 2 --abi-devices 0,0 --workload c5 --guides 40000 --batch 5000 --steps 1 --warmup 1 > "$OUT/bench_abi_c5_x2_one_gpu.json" 2> "$OUT/bench_abi_c5_x2_one_gpu.err" || true
timeout -k 10 500 python3 $ROOT/bench.py --workload c3 --steps 3 --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || true
fi
find "$OUT" -name '*kernel_trace.csv' -delete
du -sh "$OUT"
