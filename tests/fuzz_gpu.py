#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (not collected by pytest): N random configurations - genome
shapes with repeats and homopolymers, N runs, tiny contigs, read sets with near-duplicates, every
mismatch budget, optional extra PAM, both algorithms, 1-3 shards - each compared record by record with
the oracle's bit-parallel port.

    python tests/fuzz_gpu.py [N] [first_seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import varscot_amd as va  # noqa: E402
from helpers import hits_as_tuples, make_genome, mutate, random_guides, random_seq  # noqa: E402
from oracle import pyoracle  # noqa: E402


def one(ctx, seed):
    rng = np.random.default_rng(seed)
    max_mm = int(rng.integers(0, 9))
    n_guides = int(rng.choice([1, 3, 17, 64, 65, 300, 1500]))
    guides = random_guides(rng, n_guides, pam=str(rng.choice(["GG", "GG", "GA", "AG"])))
    # near-duplicate reads and low-complexity reads
    for _ in range(n_guides // 4):
        guides.append(mutate(rng, guides[int(rng.integers(0, n_guides))], int(rng.integers(0, 3))))
    if rng.integers(0, 3) == 0:
        guides += ["G" * 23, "A" * 21 + "GG", "AC" * 10 + "AGG"]
    lens = [int(x) for x in rng.choice([23, 24, 60, 500, 4000, 30000, 120000, 400000], size=int(rng.integers(1, 7)))]
    contigs = make_genome(seed, lens, guides, max_mm, n_plant=int(rng.integers(0, 400)), n_runs=int(rng.integers(0, 12)))
    # repeats: a tandem array of one planted site, a homopolymer run
    big = [i for i, n in enumerate(lens) if n >= 4000]
    if big and rng.integers(0, 2):
        c = int(rng.choice(big))
        unit = mutate(rng, guides[0], min(max_mm, 2), 0, 20) + random_seq(rng, int(rng.integers(0, 30)))
        rep = unit * int(rng.integers(2, 60))
        a = int(rng.integers(0, max(1, len(contigs[c]) - len(rep))))
        contigs[c] = (contigs[c][:a] + rep + contigs[c][a + len(rep):])[:lens[c]]
    if big and rng.integers(0, 3) == 0:
        c = int(rng.choice(big))
        a = int(rng.integers(0, len(contigs[c]) - 1200))
        contigs[c] = contigs[c][:a] + str(rng.choice(["G", "C", "A"])) * 1100 + contigs[c][a + 1100:]
    if seed % 4 == 0:
        # dense: a contig made of mutated copies of the reads (tens of thousands of hits, multi-hit blocks,
        # full token rings, hit-buffer growth)
        pieces = []
        for _ in range(int(rng.integers(2000, 12000))):
            g = guides[int(rng.integers(0, len(guides)))]
            pieces.append(mutate(rng, g, int(rng.integers(0, max_mm + 2)), 0, 21) + random_seq(rng, int(rng.integers(0, 4))))
        contigs.append("".join(pieces))
        lens.append(len(contigs[-1]))
    extra = str(rng.choice(["AG", "TT", "CC"])) if rng.integers(0, 4) == 0 else None
    want = pyoracle.search_fast(contigs, guides, max_mm, extra)
    packed = va.PackedGenome.from_sequences(contigs)
    n_rec = 0
    for algo in ("scan", "seed"):
        world = int(rng.integers(1, 4))
        # sort knobs: smaller LDS capacity / fewer bits per level force partition levels and the oversize path
        hooks = {}
        for k, choices in (("sort_cap", [0, 0, 64, 1000]), ("sort_max_bits", [0, 0, 2, 5]), ("seed_reserve", [0, 64, 1024]),
                           ("seed_shared", [-1, 0, 1, 1]), ("seed_group_out", [-1, 0, 1]), ("seed_tight", [-1, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9]),
                           ("sort_optimistic", [-1, -1, 0, 1]), ("sort_slot_cap", [0, 0, 24, 300])):
            hooks[k] = choices[int(rng.integers(0, len(choices)))]
        ctx.set_debug(**hooks)
        if world > 1 and rng.integers(0, 2):  # the shards behind the C ABI: contexts on this device, gather, merge
            m = va.MultiContext([0] * world)
            m.set_debug(**hooks)
            g = m.load_genome(packed)
            h = g.search(guides, max_mm, extra, algorithm=algo)
            got = h.to_numpy().copy()
            h.close()
            g.close()
            m.close()
            if hits_as_tuples(got) != hits_as_tuples(want):
                return "seed %d: %s multi x%d differs (m=%d, %d reads, contigs %s, extra %s, hooks %s): %d vs %d records" % (
                    seed, algo, world, max_mm, len(guides), lens, extra,
                    hooks, len(got), len(want))
            n_rec = len(got)
            continue
        parts = []
        for rank in range(world):
            b, e = packed.shard_words(rank, world)
            if e <= b:
                continue
            g = ctx.load_genome(packed, rank, world)
            h = g.search(guides, max_mm, extra, algorithm=algo)
            parts.append(h.to_numpy().copy())
            h.close()
            g.close()
        got = np.concatenate(parts) if parts else np.zeros(0, dtype=va.HIT_DTYPE)
        if world > 1:  # shards partition the positions: a stable sort on (guide, strand) merges them
            key = (got["guide"].astype(np.int64) << 1) | (got["info"] >> 31)
            got = got[np.argsort(key, kind="stable")]
        if hits_as_tuples(got) != hits_as_tuples(want):
            return "seed %d: %s x%d differs (m=%d, %d reads, contigs %s, extra %s, hooks %s): %d vs %d records" % (
                seed, algo, world, max_mm, len(guides), lens, extra,
                hooks, len(got), len(want))
        n_rec = len(got)
    # the streamed search that writes the feature rows on the way (side words through the sort), under the last hooks: records =
    # the oracle's, rows = the gathering kernel's on the same hits; sometimes through the multi-device stream as well
    if rng.integers(0, 2):
        import torch
        from varscot_amd.dist import DeviceAlias
        g = ctx.load_genome(packed)
        recs, bad_rows = [], []

        def on_batch(h, first, count, rows_dev):
            a = h.to_numpy().copy()
            recs.append(a)
            if len(a):
                got_rows = torch.as_tensor(DeviceAlias(rows_dev, 64 * len(a)), device="cuda:0").view(torch.int32).view(-1, 16).cpu().numpy().view(np.uint32).copy()
                ref, _ = h.packed_features(to_host=True)
                if not np.array_equal(got_rows, ref):
                    bad_rows.append(first)

        batch = int(rng.choice([1, 7, 64, 100, 4096]))
        g.search_streamed_rows(guides, max_mm, on_batch, batch=batch, extra_pam=extra, algorithm="seed")
        g.close()
        got = np.concatenate(recs) if recs else np.zeros(0, dtype=va.HIT_DTYPE)
        if hits_as_tuples(got) != hits_as_tuples(want) or bad_rows:
            return "seed %d: streamed rows differ (m=%d, %d reads, batch %d, contigs %s, extra %s, hooks %s): %d vs %d records, rows of batches %s" % (
                seed, max_mm, len(guides), batch, lens, extra, hooks, len(got), len(want), bad_rows[:4])
    # the classifier on the hits (score -> classify fused), the reference's forest in every node form the kernel has: same votes
    if n_rec and rng.integers(0, 3) == 0:
        from varscot_amd.classifier import Forest
        forest = Forest(os.path.join(ROOT, "varscot_amd", "models", "rfClassifier.vscrf"))
        g = ctx.load_genome(packed)
        h = g.search(guides, max_mm, extra, algorithm="seed")
        act = rng.choice([0.2, 0.45, 0.87, 0.8885, 1.03, 1.31, 1.47, 1.61, 1.99, 2.3], size=len(guides))
        votes = {}
        for form in (-1, 1, 0):
            ctx.set_debug(rf_form=form)
            votes[form], _ = forest.classify_hits(h, act)
        ctx.set_debug()
        h.close()
        g.close()
        if not (np.array_equal(votes[-1], votes[1]) and np.array_equal(votes[-1], votes[0])):
            return "seed %d: the forest's node forms vote differently (m=%d, %d reads, %d hits)" % (seed, max_mm, len(guides), n_rec)
    if rng.integers(0, 3) == 0:
        world = int(rng.integers(2, 6))
        m = va.MultiContext([0] * world)
        g = m.load_genome(packed)
        recs = []
        g.search_streamed(guides, max_mm, lambda h, first, count, votes: recs.append(h.to_numpy().copy()), batch=int(rng.choice([5, 64, 1000])),
                          extra_pam=extra, algorithm=str(rng.choice(["scan", "seed"])), score=str(rng.choice(["rows", "rows", ""])) or None)
        g.close()
        m.close()
        got = np.concatenate(recs) if recs else np.zeros(0, dtype=va.HIT_DTYPE)
        if hits_as_tuples(got) != hits_as_tuples(want):
            return "seed %d: multi stream x%d differs (m=%d, %d reads, contigs %s, extra %s): %d vs %d records" % (
                seed, world, max_mm, len(guides), lens, extra, len(got), len(want))
    return n_rec


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    pyoracle.build()
    ctx = va.Context(0)
    t0 = time.time()
    total, bad = 0, []
    for i in range(n):
        r = one(ctx, first + i)
        if isinstance(r, str):
            bad.append(r)
            print(r, flush=True)
        else:
            total += r
        if (i + 1) % 20 == 0:
            print("%d / %d configurations, %d records compared, %d failures, %.0f s" % (i + 1, n, total, len(bad), time.time() - t0),
                  flush=True)
    ctx.close()
    print("fuzz: %d configurations, %d records, %d failures" % (n, total, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
