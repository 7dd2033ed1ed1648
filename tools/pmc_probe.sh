#!/bin/bash
# pmc_probe.sh WORKLOAD [extra bench args] - run on the GPU box: SQ counters of the search kernel for one workload
set -e -o pipefail
W=${1:-c2}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/probe_$W; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d "$OUT/a" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_IFETCH SQ_WAIT_INST_LDS \
    -d "$OUT/b" -o run --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/b.err" || true
python3 - "$OUT" <<'PY'
import csv, glob, sys
acc = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        if "seed_sliced_kernel" in row["Kernel_Name"] or "seed_compare_kernel" in row["Kernel_Name"]:
            k = (row["Counter_Name"], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
    for (n, _), v in per.items():
        acc.setdefault(n, []).append(v)
for n, v in sorted(acc.items()):
    print("%-24s %.4g" % (n, sum(v) / len(v)))
PY
