#!/bin/bash
# ab_classify.sh LIB... - c5 with the classifier (two batches of 10 000 reads) under each of the given builds of the library,
# twice, alternating; prints the forest walk's time (kernels_ms.score).  Run through gpurun from the repository root.
set -o pipefail
OUT=gpurun_out/${TAG:-ab_classify}; mkdir -p "$OUT"
for rep in 1 2; do for L in "$@"; do
    VSC_LIB_PATH=$PWD/$L timeout -k 10 300 python3 bench.py --workload c5 --classify --guides 20000 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/x.json" 2> "$OUT/x.err" || { tail -5 "$OUT/x.err"; exit 1; }
    python3 - "$OUT/x.json" "$L" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "score ms", round(d["kernels_ms"]["score"], 1), "hits", d["config"]["hits_per_step"], flush=True)
PY
done; done
