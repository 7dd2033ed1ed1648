#!/usr/bin/env python3
"""Experiment: do the VALU-bound compare kernel and the HBM-bound hit sort of two independent
searches overlap when they run on two streams?  Two contexts (own stream each) search half of the c3
reads each from two host threads; throughput is compared with one context doing all reads."""
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
iters = 6

table, names = synth.contig_table(bases)
span = int(table[-1]["offset"]) + int(table[-1]["length"]) + 1
n_words = (span + 31) // 32
hi, lo, nm, _, _, _ = synth.synthetic_planes(bases, 0, n_words)
ctxs = [va.Context(0), va.Context(0)]
genomes = [va.Genome.from_shard(c, hi, lo, nm, 0, n_words, table) for c in ctxs]
for g in genomes:
    g.build_index()
ids, seqs = synth.synthetic_guides(n_guides)
codes = va.pack_guides(seqs)


def run(g, part, n, out):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        h = g.search(part, m, algorithm="seed")
        h.close()
        t.append(time.perf_counter() - t0)
    out.append(t)


# one context, all reads
o = []
run(genomes[0], codes, 2, o)
t0 = time.perf_counter()
run(genomes[0], codes, iters, o)
single = (time.perf_counter() - t0) / iters
print("single stream: %.1f ms per %d reads" % (single * 1e3, n_guides), flush=True)

for split in (2, 4):
    parts = np.array_split(codes, split)
    # every thread works through the parts in a different rotation so that phases differ
    def worker(k, out):
        g = genomes[k]
        for it in range(iters + 1):
            for j in range(split // 2):
                h = g.search(parts[(2 * j + k) % split], m, algorithm="seed")
                h.close()
            if it == 0:
                out.append(time.perf_counter())
        out.append(time.perf_counter())
    outs = [[], []]
    th = [threading.Thread(target=worker, args=(k, outs[k])) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    begin = min(o2[0] for o2 in outs)
    end = max(o2[1] for o2 in outs)
    print("two streams, %d parts: %.1f ms per %d reads" % (split, (end - begin) / iters * 1e3, n_guides), flush=True)
