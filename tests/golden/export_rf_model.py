#!/usr/bin/env python3
"""Dev-container-only: converts the reference's trained random forest
(/root/reference/VARSCOT_pipeline/classification/rfClassifier.RData, the model
classification/classificationPipeline.R:21-34 loads) into the flat file the MI355X tools read:
varscot_amd/models/rfClassifier.vscrf.  Data only (trained weights), no reference source text.

Layout (little-endian): "VSCRF001" | u32 n_trees | u32 n_nodes | u32 n_vars | n_vars x (u16 len + name)
| per tree, n_nodes entries each: i8 node_status (1 split, -1 terminal, 0 unused) | u8 best_var (1-based
predictor, 0 for terminal) | u16 left | u16 right (1-based daughters) | f64 split | u8 node_class (1-based).
Also writes tests/golden/rf_training.npz: the 696 training rows the model was fitted on (matched by
label vector against the 10 samplings of featureMatrix.RData) with the labels and the stored OOB votes.
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from rdata_reader import data_frame, load_rdata  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    rf = load_rdata("/root/reference/VARSCOT_pipeline/classification/rfClassifier.RData")["rfClassifier"]
    obj = dict(zip(rf["attr"]["names"]["val"], rf["val"]))
    fo = dict(zip(obj["forest"]["attr"]["names"]["val"], obj["forest"]["val"]))
    nt, nn = int(fo["ntree"]["val"][0]), int(fo["nrnodes"]["val"][0])
    names = obj["importance"]["attr"]["dimnames"]["val"][0]["val"]
    status = np.array(fo["nodestatus"]["val"], dtype=np.int8).reshape(nt, nn)         # [node, tree] column-major
    bestvar = np.array(fo["bestvar"]["val"], dtype=np.uint8).reshape(nt, nn)
    treemap = np.array(fo["treemap"]["val"], dtype=np.uint16).reshape(nt, 2, nn)      # [node, side, tree]
    nodepred = np.array(fo["nodepred"]["val"], dtype=np.uint8).reshape(nt, nn)
    split = np.array(fo["xbestsplit"]["val"], dtype=np.float64).reshape(nt, nn)
    assert all(int(c) == 1 for c in fo["ncat"]["val"]), "categorical splits are not supported"
    assert fo["cutoff"]["val"] == [0.5, 0.5]
    out = os.path.join(ROOT, "varscot_amd", "models", "rfClassifier.vscrf")
    with open(out, "wb") as f:
        f.write(b"VSCRF001" + struct.pack("<III", nt, nn, len(names)))
        for n in names:
            f.write(struct.pack("<H", len(n)) + n.encode())
        f.write(status.tobytes() + bestvar.tobytes() + treemap[:, 0, :].astype("<u2").tobytes() +
                treemap[:, 1, :].astype("<u2").tobytes() + split.astype("<f8").tobytes() + nodepred.tobytes())
    print(out, os.path.getsize(out), "bytes;", nt, "trees x", nn, "nodes,", len(names), "predictors")

    # the training sample: which of the ten 696-row samplings carries this label vector?
    fm = load_rdata("/root/reference/workflow/data-objects/featureMatrix.RData")["featureMatrix"]
    y = np.array(obj["y"]["val"]) - 1
    votes = np.array(obj["votes"]["val"]).reshape(2, 696).T
    found = None
    for k, d in enumerate(fm["val"]):
        f = data_frame(d)
        lab = np.array([int(v) for v in f["offtargetActivity"]])
        if np.array_equal(lab, y):
            found = (k, f)
            break
    if found is None:
        print("no sampling matches the model's label vector exactly; storing sampling 0")
        found = (0, data_frame(fm["val"][0]))
    k, f = found
    cols = list(f.keys())[1:]
    x = np.array([[float(v) for v in f[c]] for c in cols]).T
    np.savez_compressed(os.path.join(HERE, "rf_training.npz"), x=x, names=np.array(cols), y=y, oob_votes=votes, sampling=k)
    print("rf_training.npz: sampling", k, x.shape)


if __name__ == "__main__":
    main()
