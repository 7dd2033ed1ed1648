#!/bin/bash
# pmc_probe3.sh LIB - VALU/SALU instruction counts + time of the search kernel on c3 with library build LIB
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/probe3; rm -rf "$OUT"; mkdir -p "$OUT"
export VSC_LIB_PATH=$ROOT/varscot_amd/$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d "$OUT/a" -o run --output-format csv -- python3 $ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/a.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
acc = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        if "seed_sliced_kernel" in row["Kernel_Name"]:
            k = (row["Counter_Name"], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
    for (n, _), v in per.items():
        acc.setdefault(n, []).append(v)
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print({n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(acc.items())}, "search ms", d["kernels_ms"]["search"])
PY
