#!/bin/bash
# index build time (first and repeated) on the 3 Gbp genome + parity tests
mkdir -p gpurun_out/r2x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r2x/pytest_idx.log 2>&1; tail -1 gpurun_out/r2x/pytest_idx.log
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time
import varscot_amd as va
from varscot_amd import synth
ctx = va.Context(0)
packed = synth.synthetic_genome(3_000_000_000)
g = ctx.load_genome(packed)
for it in range(3):
    t = time.time(); g.build_index(extra_pam=("AG" if it % 2 else None)); w = time.time() - t
    print("build", it, "wall %.3f s" % w, "device %.1f ms" % ctx.timing()["index_ms"], "bytes", g.device_bytes)
ids, guides = synth.synthetic_guides(1000)
g.build_index()
h = g.search(guides, 6, algorithm="seed"); print("hits", len(h)); h.close()
PY
