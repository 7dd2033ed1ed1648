#!/bin/bash
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2f/pytest.log 2>&1
tail -4 gpurun_out/r2f/pytest.log
for X in 1 0; do
VSC_SORT_XCD=$X timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_c3_xcd$X.json 2> gpurun_out/r2f/bench_c3_xcd$X.err || tail -20 gpurun_out/r2f/bench_c3_xcd$X.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2f/bench_c3_xcd$X.json"))
print("xcd=$X", d["ms_per_step"], d["kernels_ms"], d["roofline_sort"]["achieved"])
PY
done
timeout -k 10 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_c2.json 2> gpurun_out/r2f/bench_c2.err || tail -20 gpurun_out/r2f/bench_c2.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2f/bench_c2.json"))
print("c2", d["ms_per_step"], d["kernels_ms"])
PY
