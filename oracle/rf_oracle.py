"""Pure-Python restatement of randomForest's classification `predict` as
classification/classificationPipeline.R:27-34 uses it (randomForest 4.6: predict.randomForest ->
C classForest / predictClassTree): per tree, descend from the root with `x[var] <= split ? left :
right` until a terminal node and take its class; type="prob" = votes / ntree; the class is the
larger vote share (cutoff 0.5/0.5), ties broken at random by R - reported here as a tie flag.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: R is not available and the reference stores only
out-of-bag votes; the third-party algorithm (randomForest, a CRAN dependency of the reference) is
restated from its published behaviour.  The forest itself is the reference's own trained model
(varscot_amd/models/rfClassifier.vscrf, exported from classification/rfClassifier.RData).
"""
import struct

import numpy as np


class Forest:
    def __init__(self, path):
        with open(path, "rb") as f:
            assert f.read(8) == b"VSCRF001"
            self.n_trees, self.n_nodes, n_vars = struct.unpack("<III", f.read(12))
            self.names = []
            for _ in range(n_vars):
                (ln,) = struct.unpack("<H", f.read(2))
                self.names.append(f.read(ln).decode())
            n = self.n_trees * self.n_nodes
            shape = (self.n_trees, self.n_nodes)
            self.status = np.frombuffer(f.read(n), dtype=np.int8).reshape(shape)
            self.best_var = np.frombuffer(f.read(n), dtype=np.uint8).reshape(shape)
            self.left = np.frombuffer(f.read(2 * n), dtype="<u2").reshape(shape)
            self.right = np.frombuffer(f.read(2 * n), dtype="<u2").reshape(shape)
            self.split = np.frombuffer(f.read(8 * n), dtype="<f8").reshape(shape)
            self.node_class = np.frombuffer(f.read(n), dtype=np.uint8).reshape(shape)

    def votes(self, row):
        """row: {predictor name: value}.  Returns the number of trees voting for class "1"."""
        x = [float(row[n]) for n in self.names]
        ones = 0
        for t in range(self.n_trees):
            k = 0
            while self.status[t, k] != -1:
                m = int(self.best_var[t, k]) - 1
                k = int(self.left[t, k]) - 1 if x[m] <= self.split[t, k] else int(self.right[t, k]) - 1
            ones += int(self.node_class[t, k]) == 2
        return ones

    def predict(self, row):
        """(probability of class "1", class 0/1, tie flag)."""
        v = self.votes(row)
        return v / self.n_trees, int(2 * v > self.n_trees), 2 * v == self.n_trees
