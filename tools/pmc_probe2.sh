#!/bin/bash
# pmc_probe2.sh WORKLOAD "COUNTER LIST" [KERNEL-NAME PART ...] - run on the GPU box: arbitrary counters of the search
# kernel (or of the kernels whose names contain one of the given parts)
set -e -o pipefail
W=${1:-c3}; CTR=${2}; shift; shift; export PMC_KERNELS="${*:-seed_sliced_kernel seed_compare_kernel}"
# TA_* (texture addresser) counters are refused: the one pass that used them on this pool (round 1, c3, together with
# SQ counters) never returned and was killed at gpurun's limit; its rocprofv3 log and the box's dmesg were lost with the
# box, so whether the profiler's serialised dispatch or one of our kernels stalled could not be established.  Until a
# run with `timeout -k 10 120` around a TA-only pass of a trivial kernel says otherwise, treat them as unsafe here.
case " $CTR " in *" TA_"*) echo "pmc_probe2.sh: TA_* counters are refused on this pool (see the comment in this script)" >&2; exit 2;; esac
ROOT=$(pwd); OUT=$ROOT/gpurun_out/probe2_$W; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR -d "$OUT/a" -o run --output-format csv -- python3 $ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/a.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, os
parts = os.environ['PMC_KERNELS'].split()
acc = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        if any(p in row["Kernel_Name"] for p in parts):
            k = (row["Counter_Name"], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
    for (n, _), v in per.items():
        acc.setdefault(n, []).append(v)
for n, v in sorted(acc.items()):
    print("%-36s %.4g" % (n, sum(v) / len(v)))
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("kernels ms", d["kernels_ms"])
PY
