"""Random-forest inference: vsc_rf_predict against the pure-Python
restatement of randomForest's predict, a statistical check against the reference's own training
labels (parity is unpinned: no R, only out-of-bag votes are stored), and the drop-in tool."""
import os
import subprocess

import numpy as np
import pytest

import varscot_amd as va
from varscot_amd.classifier import Forest, feature_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join(ROOT, "varscot_amd", "models", "rfClassifier.vscrf")


def test_model_file_and_feature_names(golden_dir):
    f = Forest(MODEL)
    assert (f.n_trees, f.n_nodes, len(f.names)) == (1000, 275, 80)
    assert "ontargetActivity" in f.names and "seedMismatches" in f.names
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    assert feature_names() == list(g["names"]) + ["ontargetActivity"]
    # every split node tests a valid column and has two daughters inside the tree
    split = f.status == 1
    assert f.feature[split].max() <= 442 and f.left[split].min() >= 1 and f.right[split].max() <= 275


@pytest.mark.gpu
def test_rf_predict_matches_restatement_and_training_labels(golden_dir):
    from oracle.rf_oracle import Forest as OracleForest
    ctx = va.Context(0)
    t = np.load(os.path.join(golden_dir, "rf_training.npz"))
    names, x, y = list(t["names"]), t["x"], t["y"]
    assert names == feature_names()
    feats = x[:, :442].astype(np.uint8)
    act = x[:, 442].astype(np.float64)
    prob, cls, tie = Forest(MODEL).predict(ctx, feats, act)
    # the three node forms of the kernel (pair nodes - the default for this forest -, compact nodes, plain nodes) vote alike
    for form in (1, 0):
        ctx.set_debug(rf_form=form)
        other = Forest(MODEL).predict(ctx, feats, act)
        assert all(np.array_equal(a, b) for a, b in zip((prob, cls, tie), other)), form
    ctx.set_debug()
    of = OracleForest(MODEL)
    for i in range(0, len(x), 5):  # every 5th row through the Python restatement: identical votes
        p, c, ti = of.predict(dict(zip(names, x[i])))
        assert prob[i] == p and cls[i] == c and bool(tie[i]) == ti
    # the forest reproduces the labels it was grown on (in-bag fit); reversing the split direction
    # would not - this is the statistical anchor of an otherwise unpinned stage
    assert (cls == y).mean() > 0.95
    # out-of-bag votes of the reference are a noisier estimate of the same quantity
    oob = t["oob_votes"][:, 1]
    assert np.corrcoef(prob, oob)[0, 1] > 0.8
    ctx.close()


@pytest.mark.gpu
def test_rf_predict_from_packed_rows_on_host_and_device():
    """vsc_rf_predict_packed decodes the columns the forest tests straight from the 64-byte packed rows - no
    442-byte expansion: same votes as the dense entry point, for rows passed from the host and for rows left in
    device memory by vsc_score_hits_packed; few rows (the trees are split over workgroups, votes meet in an
    atomic) and many rows (one workgroup walks all trees of its 512 rows)."""
    import torch
    from helpers import make_genome, random_guides
    rng = np.random.default_rng(77)
    guides = random_guides(rng, 20)
    contigs = make_genome(77, [60000, 20000], guides, 6, n_plant=300, n_runs=2)
    ctx = va.Context(0)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    h = gen.search(guides, 6)
    n = len(h)
    assert n > 200
    _, _, dense = h.scores(mit=False, features=True)
    rows, _ = h.packed_features()
    dev = torch.empty((n, 16), dtype=torch.int32, device="cuda:0")
    h.packed_features(to_host=False, dev_ptr=dev.data_ptr())
    act = rng.random(n) * 0.9 + 0.05
    forest = Forest(MODEL)
    want = forest.predict(ctx, dense, act)
    for got in (forest.predict_packed(ctx, rows, act), forest.predict_packed(ctx, n, act, dev_ptr=dev.data_ptr()),
                forest.predict_packed(ctx, rows[:3], act[:3])):
        m = len(got[0])
        assert all(np.array_equal(a[:m], b) for a, b in zip(want, got))
    assert 0 < want[1].mean() < 1 or n < 50  # both classes occur among a few hundred real hits
    big = 300_000 // n + 1  # many rows: every workgroup walks the whole forest
    many = forest.predict_packed(ctx, np.tile(rows, (big, 1)), np.tile(act, big))
    assert all(np.array_equal(np.tile(a, big), b) for a, b in zip(want, many))
    h.close()
    gen.close()
    ctx.close()


@pytest.mark.gpu
def test_fused_score_classify_equals_score_then_predict(golden_dir):
    """vsc_score_classify_hits (rows never leave the registers, 2 bytes per hit out) against the two-step path it
    replaces on a streamed search - vsc_score_hits_packed + vsc_rf_predict_packed on the same hits: identical
    votes, hit for hit; the MIT scores equal vsc_score_hits'.  Both strands, several reads with their own
    activities (values around the forest's activity thresholds), a row range, device and host destinations."""
    import torch
    from helpers import make_genome, real_guides
    names, guides, acts = real_guides(golden_dir)  # the reference's own targets and TUSCAN activities
    rng = np.random.default_rng(31)
    contigs = make_genome(31, [150000, 40000], guides, 7, n_plant=900, n_runs=2)
    ctx = va.Context(0)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    h = gen.search(guides, 7)
    rec = h.to_numpy()
    n = len(h)
    assert n > 600 and len(set(rec["guide"])) > 8
    forest = Forest(MODEL)
    act = np.array(acts, dtype=np.float64)
    act[3] = float(np.unique(forest.split[(forest.status == 1) & (forest.feature == 442)])[7])  # exactly ON a threshold (<=)
    rows, mit = h.packed_features(mit=True)
    want_prob, want_cls, want_tie = forest.predict_packed(ctx, rows, act[rec["guide"]])
    votes, mit2 = forest.classify_hits(h, act, mit=True)
    assert np.array_equal(votes / 1000.0, want_prob) and np.array_equal(2 * votes.astype(int) > 1000, want_cls.astype(bool))
    assert np.array_equal(2 * votes.astype(int) == 1000, want_tie.astype(bool)) and np.array_equal(mit2, mit)
    assert 0 < want_cls.mean() < 1
    dev = torch.zeros(n - 100, dtype=torch.int16, device="cuda:0")
    forest.classify_hits(h, act, first=37, count=n - 100, to_host=False, dev_ptr=dev.data_ptr())
    assert np.array_equal(dev.cpu().numpy().view(np.uint16), votes[37:n - 63])
    h.close()
    gen.close()
    ctx.close()


def _synthetic_forest(path, rng, n_trees, n_nodes, names, depth_first=False, split_choices=None):
    """A random forest file (the exporter's format): random binary trees grown breadth-first (randomForest's node
    numbering: the daughters of a node are the next two free numbers) over predictors of the feature matrix, with
    splits at x.5, integer splits, a few below zero and above any value, some trees a single terminal node."""
    import struct
    cols = {n: i for i, n in enumerate(feature_names())}
    status = np.zeros((n_trees, n_nodes), dtype=np.int8)
    best = np.zeros((n_trees, n_nodes), dtype=np.uint8)
    left = np.zeros((n_trees, n_nodes), dtype="<u2")
    right = np.zeros((n_trees, n_nodes), dtype="<u2")
    split = np.zeros((n_trees, n_nodes), dtype="<f8")
    cls = np.zeros((n_trees, n_nodes), dtype=np.uint8)
    for t in range(n_trees):
        size = 1 if t % 17 == 0 else min(n_nodes, int(rng.integers(3, n_nodes + 1)) | 1)  # odd: every split adds two nodes
        nxt, k = 1, 0
        while k < nxt:  # nodes in creation order; nodes behind nxt do not exist
            if nxt + 2 <= size and (rng.random() < 0.9 or k == nxt - 1):
                status[t, k] = 1
                v = int(rng.integers(0, len(names)))
                best[t, k] = v + 1
                left[t, k], right[t, k] = nxt + 1, nxt + 2
                nxt += 2
                if names[v] == "ontargetActivity":
                    split[t, k] = float(rng.choice([0.31, 0.5, 0.77, 1.02, 1.4]))
                else:
                    split[t, k] = float(rng.choice(split_choices if split_choices is not None else [0.5, 0.5, 0.5, 1.5, 2.5, 1.0, 3.0, -0.5, 300.0]))
            else:
                status[t, k] = -1
                cls[t, k] = int(rng.integers(1, 3))
            k += 1
    with open(path, "wb") as f:
        f.write(b"VSCRF001" + struct.pack("<III", n_trees, n_nodes, len(names)))
        for n in names:
            f.write(struct.pack("<H", len(n)) + n.encode())
        for a in (status, best, left, right, split, cls):
            f.write(a.tobytes())
    return cols


@pytest.mark.gpu
@pytest.mark.parametrize("n_nodes,form", [(31, -1), (511, -1), (512, -1), (512, 1), (512, 0), (31, 1), (700, -1)])
def test_rf_predict_on_synthetic_forests(tmp_path, golden_dir, n_nodes, form):
    """The node forms of the forest kernel - pair nodes (two levels per 8-byte node) and compact nodes, both walked through
    per-lane tree queues (<= 512 nodes per tree; `form` = the rf_form hook: 1 stops at the compact form, 0 at the plain one),
    and the self-looping form with wave-uniform step counts (more nodes) - against the Python restatement of
    randomForest's predict, on random forests: unbalanced trees, trees that are one terminal node, count predictors
    split at integers and at x.5, splits below zero and above every value, the activity split at a handful of
    thresholds with rows exactly on them; dense rows, packed rows and few / many rows."""
    from oracle.rf_oracle import Forest as OracleForest
    rng = np.random.default_rng(n_nodes)
    names = ["totalMismatches", "seedMismatches", "adjacentMismatches", "transitionNumber", "transversionNumber", "mismatchPos3",
             "mismatchPos17", "AtoC", "TtoG", "A1", "T20", "PAMG", "GG", "CA", "TT", "AA7", "CG19", "ontargetActivity"]
    path = str(tmp_path / "forest.vscrf")
    _synthetic_forest(path, rng, 64, n_nodes, names)
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    feats = g["feat"][::9].astype(np.uint8)
    act = rng.choice([0.2, 0.31, 0.5, 0.77, 0.9, 1.02, 1.4, 1.7], size=len(feats))
    ctx = va.Context(0)
    ctx.set_debug(rf_form=form)
    forest = Forest(path)
    prob, cls, tie = forest.predict(ctx, feats, act)
    of = OracleForest(path)
    all_names = feature_names()
    for i in range(0, len(feats), 3):
        row = dict(zip(all_names, list(feats[i]) + [act[i]]))
        p, c, ti = of.predict(row)
        assert (prob[i], cls[i], bool(tie[i])) == (p, c, ti), i
    assert 0 < cls.mean() < 1
    prob3, _, _ = forest.predict(ctx, np.tile(feats, (40, 1)), np.tile(act, 40))  # many rows: whole forest per workgroup
    assert np.array_equal(prob3, np.tile(prob, 40))
    # the other two row sources - packed rows, and rows computed in the kernel from hits - vote like the dense rows
    from helpers import make_genome, random_guides
    guides = random_guides(rng, 12)
    contigs = make_genome(n_nodes, [50000, 9000], guides, 6, n_plant=300, n_runs=1)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    h = gen.search(guides, 6)
    rec = h.to_numpy()
    assert len(rec) > 200
    _, _, dense = h.scores(mit=False, features=True)
    rows, _ = h.packed_features()
    gact = rng.choice([0.2, 0.31, 0.5, 0.77, 1.02, 1.4, 1.7], size=len(guides))
    want = forest.predict(ctx, dense, gact[rec["guide"]])
    got = forest.predict_packed(ctx, rows, gact[rec["guide"]])
    assert all(np.array_equal(x, y) for x, y in zip(want, got))
    votes, _ = forest.classify_hits(h, gact)
    assert np.array_equal(votes / float(forest.n_trees), want[0])
    h.close()
    gen.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_rf_node_forms_agree_on_random_forests(tmp_path, golden_dir, seed):
    """Random forests of random shapes through every node form the kernel has (rf_form hook: -1 the best the forest allows,
    1 at most compact nodes, 0 plain nodes): identical votes, and the Python restatement's on a sample.  The shapes cover the
    limits of the pair form - more than 256 distinct tests (count predictors split at 0 .. 18), more than 127 pair nodes per
    tree (deep trees of 511 nodes) - where the library has to fall back, single-node trees and one-tree forests."""
    from oracle.rf_oracle import Forest as OracleForest
    rng = np.random.default_rng(1000 + seed)
    counts = ["AA", "AC", "AG", "AT", "CA", "CC", "CG", "CT", "GA", "GC", "GG", "GT", "TA", "TC", "TG", "TT", "totalMismatches",
              "seedMismatches", "transitionNumber", "transversionNumber", "adjacentMismatches"]
    flags = ["mismatchPos%d" % i for i in range(1, 22)] + ["A1", "C8", "G16", "T20", "PAMA", "PAMG", "AC4", "GC16", "TT19", "AtoC", "GtoA", "TtoG"]
    many_tests = seed % 3 == 0   # > 256 (predictor, threshold) pairs: no pair form
    deep = seed % 3 == 1         # > 127 pair nodes per tree: no pair form
    names = (counts if many_tests else counts[:6] + flags) + ["ontargetActivity"]
    n_nodes = 511 if deep else 255 if many_tests else int(rng.choice([3, 9, 41, 121, 255]))
    n_trees = 150 if many_tests else int(rng.choice([2, 7, 64, 150]))
    path = str(tmp_path / "forest.vscrf")
    _synthetic_forest(path, rng, n_trees, n_nodes, names,
                      split_choices=[float(v) + float(h) for v in range(-1, 19) for h in (0.0, 0.5)] if many_tests else None)
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    feats = g["feat"][seed::7].astype(np.uint8)
    act = rng.choice([0.2, 0.31, 0.5, 0.77, 0.9, 1.02, 1.4, 1.7], size=len(feats))
    ctx = va.Context(0)
    votes = {}
    for form in (-1, 1, 0):
        ctx.set_debug(rf_form=form)
        votes[form] = Forest(path).predict(ctx, feats, act)
    for form in (1, 0):
        assert all(np.array_equal(a, b) for a, b in zip(votes[-1], votes[form])), form
    of = OracleForest(path)
    all_names = feature_names()
    for i in range(0, len(feats), 41):
        p, c, ti = of.predict(dict(zip(all_names, list(feats[i]) + [act[i]])))
        assert (votes[-1][0][i], votes[-1][1][i], bool(votes[-1][2][i])) == (p, c, ti), i
    ctx.close()


@pytest.mark.gpu
def test_classification_pipeline_tool(tmp_path, golden_dir):
    """TSV + feature file in, TSV with the Score column replaced out (classificationPipeline.R:36-48)."""
    from oracle.rf_oracle import Forest as OracleForest
    t = np.load(os.path.join(golden_dir, "rf_training.npz"))
    names, x = list(t["names"]), t["x"][:40]
    with open(tmp_path / "feat.txt", "w") as f:
        f.write("\t".join(names) + "\n")
        for i, row in enumerate(x):
            f.write("t_%d\t" % (i + 1) + "".join("%d\t" % v for v in row[:442]) + "%g\n" % row[442])
    tsv_rows = ["chr1\t%d\t%d\tt_%d\t.\t+\tACGTACGTACGTACGTACGTAGG\t2\t3,7\tREF" % (100 * i, 100 * i + 23, i + 1) for i in range(40)]
    header = "#Chr\tStart\tEnd\tTargetsite\tScore\tStrand\tSequence\tMismatch_Number\tMismatch_Positions\tVariants"
    of = OracleForest(MODEL)
    for flag in ("TRUE", "FALSE"):
        (tmp_path / "out.txt").write_text("\n".join([header] + tsv_rows) + "\n")
        r = subprocess.run([os.path.join(ROOT, "varscot_amd", "bin", "classification_pipeline"), str(tmp_path / "out.txt"),
                            str(tmp_path / "feat.txt"), flag], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = (tmp_path / "out.txt").read_text().splitlines()
        assert lines[0] == header.replace("Targetsite", "Name")  # the R script renames the column (:39,43)
        for i, line in enumerate(lines[1:]):
            row = dict(zip(names, [float("%g" % v) if k == 442 else v for k, v in enumerate(x[i])]))
            p, c, _ = of.predict(row)
            want = ("%.15g" % p) if flag == "TRUE" else str(c)
            assert line.split("\t")[4] == want
            assert line.split("\t")[:4] == tsv_rows[i].split("\t")[:4]
    assert subprocess.run([os.path.join(ROOT, "varscot_amd", "bin", "classification_pipeline")], capture_output=True).returncode == 1
