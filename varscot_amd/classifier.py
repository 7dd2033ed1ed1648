"""Random-forest classifier of VARSCOT (classification/classificationPipeline.R) on the GPU:
loads the exported model (varscot_amd/models/rfClassifier.vscrf) and runs vsc_rf_predict."""
import ctypes as C
import os
import struct

import numpy as np

from . import _lib
from ._lib import N_FEATURES, RfModel, check, lib, ptr

DEFAULT_MODEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "rfClassifier.vscrf")


def feature_names():
    """The 443 column names of the feature file (variant_processing/feature_matrix.h:155-203)."""
    n = ["totalMismatches"] + ["mismatchPos%d" % i for i in range(1, 22)]
    n += ["AtoC", "AtoG", "AtoT", "CtoA", "CtoG", "CtoT", "GtoA", "GtoC", "GtoT", "TtoA", "TtoC", "TtoG"]
    n += ["transitionNumber", "transversionNumber"]
    n += [b + str(i) for i in range(1, 21) for b in "ACGT"] + ["PAMA", "PAMC", "PAMG", "PAMT"]
    pairs = [a + b for a in "ACGT" for b in "ACGT"]
    n += [p + str(i) for i in range(1, 20) for p in pairs] + pairs
    return n + ["adjacentMismatches", "seedMismatches", "ontargetActivity"]


class Forest:
    def __init__(self, path=DEFAULT_MODEL):
        with open(path, "rb") as f:
            if f.read(8) != b"VSCRF001":
                raise ValueError("%s is not a VARSCOT forest file" % path)
            self.n_trees, self.n_nodes, n_vars = struct.unpack("<III", f.read(12))
            self.names = []
            for _ in range(n_vars):
                (ln,) = struct.unpack("<H", f.read(2))
                self.names.append(f.read(ln).decode())
            n = self.n_trees * self.n_nodes
            self.status = np.frombuffer(f.read(n), dtype=np.int8).copy()
            best_var = np.frombuffer(f.read(n), dtype=np.uint8).copy()
            self.left = np.frombuffer(f.read(2 * n), dtype="<u2").copy()
            self.right = np.frombuffer(f.read(2 * n), dtype="<u2").copy()
            self.split = np.frombuffer(f.read(8 * n), dtype="<f8").copy()
            self.node_class = np.frombuffer(f.read(n), dtype=np.uint8).copy()
        cols = {name: i for i, name in enumerate(feature_names())}
        col_of_var = np.array([cols[v] for v in self.names], dtype=np.uint16)
        self.feature = np.zeros(n, dtype=np.uint16)
        split_nodes = best_var > 0
        self.feature[split_nodes] = col_of_var[best_var[split_nodes].astype(np.int64) - 1]

    def predict(self, ctx, features, activity):
        """features: uint8[n, 442]; activity: float64[n].  Returns (prob, class, tie)."""
        features = np.ascontiguousarray(features, dtype=np.uint8).reshape(-1, N_FEATURES)
        activity = np.ascontiguousarray(activity, dtype=np.float64)
        n = len(features)
        assert len(activity) == n
        m = RfModel(self.n_trees, self.n_nodes, ptr(self.status), ptr(self.feature), ptr(self.left), ptr(self.right),
                    ptr(self.split), ptr(self.node_class))
        prob = np.empty(n, dtype=np.float64)
        cls = np.empty(n, dtype=np.uint8)
        tie = np.empty(n, dtype=np.uint8)
        check(lib().vsc_rf_predict(ctx._h, C.byref(m), ptr(features), ptr(activity), n, ptr(prob), ptr(cls), ptr(tie)), ctx._h)
        return prob, cls, tie

    def predict_packed(self, ctx, rows, activity, dev_ptr=None):
        """The same from packed 64-byte rows: uint32[n, 16] on the host, or (dev_ptr: device address, rows = row
        count) in device memory as vsc_score_hits_packed left them.  Returns (prob, class, tie)."""
        activity = np.ascontiguousarray(activity, dtype=np.float64)
        n = len(activity)
        if dev_ptr is None:
            rows = np.ascontiguousarray(rows, dtype=np.uint32).reshape(-1, 16)
            assert len(rows) == n
        m = RfModel(self.n_trees, self.n_nodes, ptr(self.status), ptr(self.feature), ptr(self.left), ptr(self.right),
                    ptr(self.split), ptr(self.node_class))
        prob = np.empty(n, dtype=np.float64)
        cls = np.empty(n, dtype=np.uint8)
        tie = np.empty(n, dtype=np.uint8)
        src = C.c_void_p(dev_ptr) if dev_ptr is not None else ptr(rows)
        check(lib().vsc_rf_predict_packed(ctx._h, C.byref(m), src, int(dev_ptr is not None), ptr(activity), n, ptr(prob),
                                          ptr(cls), ptr(tie)), ctx._h)
        return prob, cls, tie

    def classify_hits(self, hits, guide_activity, first=0, count=None, to_host=True, mit=False, dev_ptr=None):
        """vsc_score_classify_hits: score -> classify fused - the hits' feature rows are computed in registers and
        walked through the forest, 2 bytes (the votes for class "1") per hit leave the kernel.
        guide_activity: float64 per read of the search.  Returns (votes uint16[count] | None, mit | None);
        prob = votes / n_trees, class = 2 * votes > n_trees, tie = 2 * votes == n_trees."""
        count = len(hits) - first if count is None else count
        act = np.ascontiguousarray(guide_activity, dtype=np.float64)
        assert len(act) == len(hits.codes)
        m = RfModel(self.n_trees, self.n_nodes, ptr(self.status), ptr(self.feature), ptr(self.left), ptr(self.right),
                    ptr(self.split), ptr(self.node_class))
        votes = np.empty(count, dtype=np.uint16) if to_host else None
        ms = np.empty(count, dtype=np.float64) if mit else None
        ctx = hits.genome.ctx
        check(lib().vsc_score_classify_hits(ctx._h, hits.genome._h, hits._h, ptr(hits.codes), len(hits.codes), ptr(act), C.byref(m),
                                            first, count, C.c_void_p(dev_ptr) if dev_ptr else None, ptr(votes), ptr(ms)), ctx._h)
        return votes, ms
