#!/bin/bash
# finalize phase stamps from an instrumented build (built out of tree into varscot_amd/libvarscot_hip_stamps.so)
mkdir -p gpurun_out/r2x
VSC_LIB_PATH=$PWD/varscot_amd/libvarscot_hip_stamps.so timeout -k 10 600 python - > gpurun_out/r2x/stamps.log 2>&1 <<'PY'
import ctypes, numpy as np
import varscot_amd as va
from varscot_amd import synth, _lib
ctx = va.Context(0)
packed = synth.synthetic_genome(3_000_000_000)
ids, guides = synth.synthetic_guides(10_000)
g = ctx.load_genome(packed); g.build_index()
lib = _lib.lib() if callable(getattr(_lib, "lib", None)) else ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 10)()
for it in range(3):
    h = g.search(guides, 8, algorithm="seed"); h.close()
    lib.vsc_debug_fin_stamps(out)
    v = list(out)
    bins = max(1, v[9])
    print("bins", v[9], "per-bin cycles", [round(x / bins) for x in v[:5]], "timing", {k: round(x, 2) for k, x in ctx.timing().items() if k.endswith("_ms")})
PY
tail -5 gpurun_out/r2x/stamps.log
