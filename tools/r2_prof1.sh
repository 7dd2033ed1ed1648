#!/bin/bash
# kernel statistics + SQ counters of one c3 bench run (separate passes)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2e; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
    -d "$OUT/pmc_sq" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
echo "pmc rc=$?"
for f in "$OUT"/pmc_*/*counter_collection.csv "$OUT"/pmc_*/*/*counter_collection.csv; do
    [ -f "$f" ] || continue
    { head -1 "$f"; grep -E 'vsc::' "$f" || true; } > "$f.small" && mv "$f.small" "$f"
done
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*kernel_stats.csv' | head -1 | xargs cat | cut -c1-200 | head -20
python3 - "$OUT" <<'PY'
import csv, glob, sys
acc = {}
for path in glob.glob(sys.argv[1] + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0][:50], row["Counter_Name"])
        acc.setdefault(k, [0.0, set()])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1].add(row["Dispatch_Id"])
for (kn, cn), (v, d) in sorted(acc.items()):
    print("%-52s %-24s %.4g per dispatch (%d)" % (kn, cn, v / len(d), len(d)))
PY
