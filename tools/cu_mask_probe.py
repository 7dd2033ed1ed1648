#!/usr/bin/env python3
"""Experiment: how do the kernels of a c3 step scale with the compute units they may use?  The search kernel is bound by
instruction issue, the sort kernels by HBM: if the sort reached its rate on a fraction of the CUs, searching one part of the
reads on the other CUs at the same time would shorten the step.  One context per CU mask (hipExtStreamCreateWithCUMask, every
k-th CU so that all XCDs take part), the library's own phase times; then two contexts with complementary masks, half of the
reads each, from two host threads.   python3 tools/cu_mask_probe.py [bases] [reads] [mismatches]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import varscot_amd as va  # noqa: E402
from varscot_amd import synth  # noqa: E402

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
n_guides = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N_CUS = 256

table, names = synth.contig_table(bases)
span = int(table[-1]["offset"]) + int(table[-1]["length"]) + 1
n_words = (span + 31) // 32
hi, lo, nm, _, _, _ = synth.synthetic_planes(bases, 0, n_words)
ids, seqs = synth.synthetic_guides(n_guides)
codes = va.pack_guides(seqs)


def mask_of(pred):
    w = np.zeros(N_CUS // 32, dtype=np.uint32)
    for cu in range(N_CUS):
        if pred(cu):
            w[cu // 32] |= np.uint32(1 << (cu % 32))
    return w


def timed(genome, ctx, part, reps=3):
    acc = {}
    h = genome.search(part, m, algorithm="seed")
    h.close()
    t0 = time.perf_counter()
    for _ in range(reps):
        h = genome.search(part, m, algorithm="seed")
        t = ctx.timing()
        for k in ("scan_ms", "prep_ms", "sort_ms", "finalize_ms"):
            acc[k] = acc.get(k, 0.0) + t[k] / reps
        h.close()
    acc["wall_ms"] = (time.perf_counter() - t0) / reps * 1e3
    return acc


masks = [("all 256 CUs", None),
         ("3 of 4 CUs (cu % 4 != 3)", mask_of(lambda c: c % 4 != 3)),
         ("1 of 2 CUs (cu % 2 == 0)", mask_of(lambda c: c % 2 == 0)),
         ("1 of 4 CUs (cu % 4 == 3)", mask_of(lambda c: c % 4 == 3)),
         ("the first 192 CUs", mask_of(lambda c: c < 192)),
         ("the last 64 CUs", mask_of(lambda c: c >= 192))]
for name, mask in masks:
    ctx = va.Context(0, cu_mask=mask)
    g = va.Genome.from_shard(ctx, hi, lo, nm, 0, n_words, table)
    g.build_index()
    r = timed(g, ctx, codes)
    print("%-28s search %.2f prep %.2f partition %.2f finalize %.2f wall %.2f ms" %
          (name, r["scan_ms"], r["prep_ms"], r["sort_ms"], r["finalize_ms"], r["wall_ms"]), flush=True)
    g.close()
    ctx.close()

# two contexts on complementary parts of the device, half of the reads each, at the same time
for name, pa, pb in (("3 of 4 | 1 of 4", lambda c: c % 4 != 3, lambda c: c % 4 == 3), ("1 of 2 | 1 of 2", lambda c: c % 2 == 0, lambda c: c % 2 == 1)):
    ctxs = [va.Context(0, cu_mask=mask_of(pa)), va.Context(0, cu_mask=mask_of(pb))]
    gens = [va.Genome.from_shard(c, hi, lo, nm, 0, n_words, table) for c in ctxs]
    for g in gens:
        g.build_index()
    halves = np.array_split(codes, 2)
    iters = 4

    def worker(k):
        for it in range(iters + 1):
            h = gens[k].search(halves[k], m, algorithm="seed")
            h.close()
            if it == 0:
                barrier.wait()

    barrier = threading.Barrier(3)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    print("two contexts, %s, half of the reads each: %.2f ms per %d reads" % (name, (time.perf_counter() - t0) / iters * 1e3, n_guides), flush=True)
    for g in gens:
        g.close()
    for c in ctxs:
        c.close()
