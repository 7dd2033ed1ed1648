#!/bin/bash
# forest kernel variants (tools/build_variant.sh) on one c5 batch with the classifier on the path
mkdir -p gpurun_out/rfsweep
for v in varscot_hip vsc_rfc3 vsc_rf512c2 vsc_rf512c3 vsc_rf1024c4; do
  VSC_LIB_PATH=$PWD/varscot_amd/lib$v.so timeout -k 10 300 python3 bench.py --workload c5 --guides 10000 --steps 1 --warmup 1 --no-cpu-baseline --classify > gpurun_out/rfsweep/$v.json 2> gpurun_out/rfsweep/$v.err || { tail -3 gpurun_out/rfsweep/$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/rfsweep/$v.json').read().strip().splitlines()[-1]); print('$v score ms', round(d['kernels_ms']['score'],1))"
done
