"""What ONE shard of an N-way genome-sharded c3 search costs on one GPU (kernel times of vsc_timing and the wall time
of the call): python tools/shard_probe.py on the GPU box.  DESIGN.md section 5 quotes its numbers."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import varscot_amd as va
from varscot_amd import synth
packed = synth.synthetic_genome(3_000_000_000)
ids, guides = synth.synthetic_guides(10000)
guides = va.pack_guides(guides) if not isinstance(guides, np.ndarray) else guides  # packed once: the wall time below is the call's
ctx = va.Context(0)
for world, rank in ((8, 0), (8, 3), (8, 7), (4, 1), (1, 0)):
    g = ctx.load_genome(packed, rank, world)
    for rep in range(3):
        t0 = time.perf_counter()
        h = g.search(guides, 8, algorithm="seed")
        dt = time.perf_counter() - t0
        t = ctx.timing()
        n = h.n if hasattr(h, "n") else len(h)
        h.close()
    print(world, rank, "wall ms", round(dt * 1e3, 2), {k: (round(t[k], 2) if isinstance(t[k], float) else t[k]) for k in ("scan_ms", "prep_ms", "sort_ms", "finalize_ms", "total_ms", "hits", "pairs", "passes", "sort_levels", "sort_fallbacks", "seed_cut", "list_entries")}, flush=True)
    g.close()
    ctx.release_scratch() if hasattr(ctx, "release_scratch") else None
