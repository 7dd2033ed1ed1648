#!/bin/bash
# packed scoring of a c3 result with parts of the kernel removed (variant libraries built out of tree)
mkdir -p gpurun_out/r2x
for v in 0 1 2 3 4; do
VSC_LIB_PATH=$PWD/varscot_amd/libvsc_var$v.so timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch
import varscot_amd as va
from varscot_amd import synth
ctx = va.Context(0)
packed = synth.synthetic_genome(3_000_000_000)
ids, guides = synth.synthetic_guides(5_000)
g = ctx.load_genome(packed); g.build_index()
h = g.search(guides, 8, algorithm="seed")
n = len(h)
rows = torch.empty((n, 16), dtype=torch.int32, device="cuda:0")
ts = []
for it in range(3):
    h.packed_features(to_host=False, dev_ptr=rows.data_ptr())
    ts.append(round(ctx.timing()["score_ms"], 2))
print(os.environ["VSC_LIB_PATH"][-8:], "rows", n, "score_ms", ts)
PY
done
