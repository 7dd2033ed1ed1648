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
