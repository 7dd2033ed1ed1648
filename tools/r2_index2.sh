#!/bin/bash
# first index build in a fresh process, committed library against the working tree's, alternating
for rep in 1 2 3; do for v in head tree; do
LIB=$PWD/varscot_amd/libvsc_phead.so; [ $v == tree ] && LIB=$PWD/varscot_amd/libvarscot_hip.so
VSC_LIB_PATH=$LIB timeout -k 10 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import time
import varscot_amd as va
from varscot_amd import synth
ctx = va.Context(0)
packed = synth.synthetic_genome(3_000_000_000)
g = ctx.load_genome(packed)
t = time.time(); g.build_index(); w = time.time() - t
print("$v", "wall %.3f s" % w, "device %.1f ms" % ctx.timing()["index_ms"])
PY
done; done
