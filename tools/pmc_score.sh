#!/bin/bash
# pmc_score.sh - counters of score_packed_kernel on two 5 000-read batches of c5
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/probe_score; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for G in "SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
rocprofv3 --pmc $G -d "$OUT/$(echo $G | cut -c1-8)" -o run --output-format csv -- python3 $ROOT/bench.py --workload c5 --guides 10000 --batch 5000 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$OUT/err.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
acc = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        if "score_packed_kernel" in row["Kernel_Name"]:
            k = (row["Counter_Name"], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
    for (n, _), v in per.items():
        acc.setdefault(n, []).append(v)
print({n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(acc.items())})
PY
