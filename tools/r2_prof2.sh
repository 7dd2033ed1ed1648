#!/bin/bash
# c3: kernel statistics, SQ counters, FETCH_SIZE, WRITE_SIZE (one pass each)
set -o pipefail
TAG=${1:-r2k}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD \
    -d "$OUT/pmc_sq" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_sq.err"; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/pmc_write.err"; echo "write rc=$?"
for f in "$OUT"/pmc_*/*counter_collection.csv "$OUT"/pmc_*/*/*counter_collection.csv; do
    [ -f "$f" ] || continue
    { head -1 "$f"; grep -E 'vsc::' "$f" || true; } > "$f.small" && mv "$f.small" "$f"
done
find "$OUT" -name '*kernel_trace.csv' -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys
acc = {}
for path in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(x in kn for x in ("bin_", "seed_sliced")): continue
        k = (kn[:40], row["Counter_Name"])
        acc.setdefault(k, [0.0, set()])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1].add(row["Dispatch_Id"])
for (kn, cn), (v, d) in sorted(acc.items()):
    print("%-42s %-20s %.4g" % (kn, cn, v / len(d)))
PY
cut -d, -f1-4 "$OUT"/stats/*kernel_stats.csv | grep -E "vsc::" | head -12
