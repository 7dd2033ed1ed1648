#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz|*.tsv from the reference's own data fixtures.

Run in the dev container only (needs /root/reference, which never travels):

    python tests/golden/make_feature_golden.py

Outputs (data only - inputs and expected outputs, no reference source text):

  features_golden.npz
      on[6960], off[6960]        23-mers  (workflow/data-objects/datasetsSampling.RData,
                                  built by workflow/processDataForModel.R:378-394)
      feat[6960, 442] uint8      the 442 sequence features of every pair, in the column order
                                  of variant_processing/feature_matrix.h:155-202
                                  (workflow/data-objects/featureMatrix.RData, built by
                                  workflow/evalFunctions.R:7-126 via classificationModel.R:21-23)
      names[442]                 column names as stored in the RData
      activity[6960] float64     the ontargetActivity column (col 442)
      nm[6960] uint8, cls[6960] uint8, chrom[6960], start[6960] int64, target[6960]
                                  the NM, Class, Chr, Start and Targetsite columns of datasetsSampling.
                                  On the 3480 Class-0 rows NM is the `NM:i` tag of VARSCOT's OWN SAM
                                  output (workflow/processDataForModel.R:257-258 reads
                                  guideseq-data/bidir_guideseq.sam, :284 takes field 3 of the tag,
                                  :378-394 samples the records): (guide, site, NM) triples the
                                  reference's mapper emitted at <= 8 mismatches - the one reference-held
                                  output of the search.  Class-1 rows are GUIDE-seq sites (NM from the
                                  GUIDE-seq tables), not mapper output - but see mapper_row.
      mapper_row[6960] int64     Class-1 rows: the row of VARSCOT's SAM output that holds this GUIDE-seq site
                                  (workflow/data-objects/indexGuideSeq.RData, built by processDataForModel.R:262-279:
                                  match on Targetsite, Chr, Start; -1 = not reported).  All 348 are >= 1: the
                                  reference's mapper reported every one of these sites.  Class-0 rows: 0.
  siteseq_pairs.tsv
      4443 (on, off, NM, strand) pairs incl. non-GG PAMs, NM up to 14
      (workflow/data-objects/offtargetBiochemicalData.RData) - inputs only.
  crispor_mit.tsv
      (guide, offtarget, mitOfftargetScore) rows of
      workflow/pipeline-comparison/crispor-siteseq-offtargets.txt for which CRISPOR's
      formula coincides with variant_processing/mit_score.h:12-68 (fewer than two
      mismatches in the 20-mer, or sum of consecutive distances divisible by their count;
      CRISPOR floors the mean, VARSCOT does not - SURVEY.md section 4).
  guideseq_sam_rows.tsv
      the 348 GUIDE-seq sites of datasetsSampling (Class 1) with the strand the GUIDE-seq table gives them
      (workflow/guideseq-data/datasetGUIDESeq.xlsx, column Strand; joined on Targetsite, chromosome, Start) and the
      row of VARSCOT's SAM output the reference found them in (mapper_row above): target, chrom, start (0-based),
      strand, nm, sam_row.  Row numbers of known records = what is left of the ORDER of the reference's output.
  guides_ontargets.tsv
      the 9 GUIDE-seq + 7 SITE-seq on-target 23-mers with their activity values
      (workflow/guideseq-data/guideseqOntargets.fasta, guideseqOntargetActivity.txt,
       workflow/siteseq-data/siteseqOntargets.fasta, siteseqOntargetActivity.txt).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from rdata_reader import data_frame, load_rdata  # noqa: E402

REF = "/root/reference/workflow"


def main():
    ds = load_rdata(f"{REF}/data-objects/datasetsSampling.RData")["datasetsSampling"]
    found = [int(v) for v in load_rdata(f"{REF}/data-objects/indexGuideSeq.RData")["indexGuideSeq"]["val"]]
    fm = load_rdata(f"{REF}/data-objects/featureMatrix.RData")["featureMatrix"]
    on, off, feat, act = [], [], [], []
    nm, cls, chrom, start, target, mapper_row = [], [], [], [], [], []
    names = None
    for d, f in zip(ds["val"], fm["val"]):
        d = data_frame(d)
        f = data_frame(f)
        cols = list(f.keys())
        assert cols[0] == "offtargetActivity" and cols[-1] == "ontargetActivity" and len(cols) == 444
        if names is None:
            names = cols[1:-1]
        assert names == cols[1:-1]
        n = len(d["Target_Sequence"])
        on += d["Target_Sequence"]
        off += d["Offtarget_Sequence"]
        nm += [int(v) for v in d["NM"]]
        cls += [int(v) for v in d["Class"]]
        chrom += d["Chr"]
        start += [int(v) for v in d["Start"]]
        target += d["Targetsite"]
        n1 = sum(1 for v in d["Class"] if int(v) == 1)
        assert n1 == len(found) and all(int(v) == 1 for v in d["Class"][:n1])  # the GUIDE-seq rows come first, in `data` order
        mapper_row += found + [0] * (n - n1)
        m = np.zeros((n, 442), dtype=np.uint8)
        for j, c in enumerate(names):
            m[:, j] = np.asarray([int(float(v)) for v in f[c]], dtype=np.uint8)
        feat.append(m)
        act += [float(v) for v in f["ontargetActivity"]]
    feat = np.concatenate(feat)
    assert feat.shape == (6960, 442), feat.shape
    np.savez_compressed(f"{HERE}/features_golden.npz", on=np.array(on), off=np.array(off), feat=feat,
                        names=np.array(names), activity=np.array(act), nm=np.array(nm, dtype=np.uint8),
                        cls=np.array(cls, dtype=np.uint8), chrom=np.array(chrom), start=np.array(start, dtype=np.int64),
                        target=np.array(target), mapper_row=np.array(mapper_row, dtype=np.int64))
    print("features_golden.npz", feat.shape)

    # strands of the GUIDE-seq sites: the xlsx is a zip of XML sheets (shared strings + one sheet)
    import re
    import zipfile
    import xml.etree.ElementTree as ET
    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}
    with zipfile.ZipFile(f"{REF}/guideseq-data/datasetGUIDESeq.xlsx") as z:
        shared = ["".join(t.itertext()) for t in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", ns)]
        sheet = ET.fromstring(z.read("xl/worksheets/sheet1.xml"))
    table = []
    for r in sheet.find("m:sheetData", ns).findall("m:row", ns):
        cells = {}
        for c in r.findall("m:c", ns):
            v = c.find("m:v", ns)
            if v is not None:
                cells[re.match(r"[A-Z]+", c.get("r")).group(0)] = shared[int(v.text)] if c.get("t") == "s" else v.text
        table.append(cells)
    assert table[0]["A"] == "#Chromosome" and table[0]["F"] == "Strand" and table[0]["H"] == "Targetsite"
    strand_of = {(c["H"], c["A"], int(c["B"])): c["F"] for c in table[1:]}
    n1 = len(found)
    with open(f"{HERE}/guideseq_sam_rows.tsv", "w") as out:
        out.write("#target\tchrom\tstart\tstrand\tnm\tsam_row\n")
        for i in range(n1):  # the Class-1 rows are the same in all ten datasets: the first one's
            key = (str(target[i]), str(chrom[i]), int(start[i]) - 1)  # (the R script works with 1-based starts)
            out.write("%s\t%s\t%d\t%s\t%d\t%d\n" % (key[0], key[1], key[2], strand_of[key], nm[i], mapper_row[i]))
    print("guideseq_sam_rows.tsv", n1)

    bd = data_frame(load_rdata(f"{REF}/data-objects/offtargetBiochemicalData.RData")["offtargetBiochemicalData"])
    with open(f"{HERE}/siteseq_pairs.tsv", "w") as out:
        out.write("#on\toff\tNM\tstrand\n")
        for a, b, nm, s in zip(bd["Target_Sequence"], bd["Offtarget_Sequence"], bd["NM"], bd["Strand"]):
            out.write(f"{a}\t{b}\t{int(nm)}\t{s}\n")
    print("siteseq_pairs.tsv", len(bd["Chr"]))

    kept = 0
    with open(f"{REF}/pipeline-comparison/crispor-siteseq-offtargets.txt") as f, \
            open(f"{HERE}/crispor_mit.tsv", "w") as out:
        out.write("#guide\tofftarget\tmitOfftargetScore\n")
        next(f)
        seen = set()
        for line in f:
            p = line.rstrip("\n").split("\t")
            g, o, score = p[1], p[2], p[4]
            if len(g) != 23 or len(o) != 23 or (g, o) in seen:
                continue
            pos = [i for i in range(20) if g[i] != o[i]]
            if len(pos) >= 2 and (pos[-1] - pos[0]) % (len(pos) - 1) != 0:
                continue
            seen.add((g, o))
            out.write(f"{g}\t{o}\t{score}\n")
            kept += 1
    print("crispor_mit.tsv", kept)

    rows = []
    for fa, actf in (("guideseq-data/guideseqOntargets.fasta", "guideseq-data/guideseqOntargetActivity.txt"),
                     ("siteseq-data/siteseqOntargets.fasta", "siteseq-data/siteseqOntargetActivity.txt")):
        acts = {}
        with open(f"{REF}/{actf}") as f:
            for line in f:
                p = line.split()
                if len(p) >= 3:
                    try:
                        acts[p[0]] = float(p[2])
                    except ValueError:
                        pass
        with open(f"{REF}/{fa}") as f:
            name = None
            for line in f:
                line = line.strip()
                if line.startswith(">"):
                    name = line[1:]
                elif line:
                    rows.append((name, line, acts.get(name)))
    with open(f"{HERE}/guides_ontargets.tsv", "w") as out:
        out.write("#name\tsequence\tactivity\n")
        for r in rows:
            out.write("%s\t%s\t%r\n" % r)
    print("guides_ontargets.tsv", len(rows))


if __name__ == "__main__":
    main()
