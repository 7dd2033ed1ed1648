"""The driver's two routes on the CPU: `driver/VARSCOT` run over a directory of tools in which every program that needs the
device is built with the host stand-ins of tools/multi_tsan (a brute-force search over the small test genome, invented
scores and votes - see its README; the device-free tools bidir_index, vcf_loader, fasta_writer are the shipped ones).  The
one-process route (varscot_pipeline: records -> potential off-targets, windows straight from the packed planes, merge,
scores, text) must write the files of the staged route (bidir_mapping's SAM text -> bam_merger[_ref_only] ->
classification_pipeline -> sort) byte for byte - what tests/test_pipeline.py holds on the GPU box with the real library; here
the HOST code of both routes is what is compared, the hits and scores are the stand-ins' on both sides."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from test_pipeline import BIN, build_scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "varscot_amd", "csrc")
STUBS = os.path.join(ROOT, "tools", "multi_tsan")


@pytest.fixture(scope="module")
def stand_in_bin(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    d = tmp_path_factory.mktemp("stand_in_bin")
    for tool in ("bidir_index", "vcf_loader", "fasta_writer"):  # the shipped programs (they find the library relative to themselves)
        (d / tool).write_text('#!/bin/bash\nexec "%s" "$@"\n' % os.path.join(BIN, tool))
        os.chmod(d / tool, 0o755)
    flags = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + STUBS]
    scores, search, pack = (os.path.join(STUBS, "stub_scores.cpp"), os.path.join(STUBS, "stub_search.cpp"), os.path.join(CSRC, "vsc_pack.cpp"))
    builds = {"bam_merger_ref_only": [scores, pack], "bam_merger": [scores, pack], "classification_pipeline": [scores, pack],
              "bidir_mapping": [scores, search, pack, os.path.join(CSRC, "vsc_windows.cpp")],
              "varscot_pipeline": [scores, search, pack, os.path.join(CSRC, "vsc_windows.cpp")]}

    def build(item):
        tool, extra = item
        return tool, subprocess.run(flags + [os.path.join(CSRC, "tools", tool + ".cpp")] + extra + ["-pthread", "-o", str(d / tool)],
                                    capture_output=True, text=True, timeout=900)
    with ThreadPoolExecutor(max_workers=5) as pool:
        for tool, b in pool.map(build, builds.items()):
            assert b.returncode == 0, tool + ": " + b.stderr[-2000:]
    return d


@pytest.mark.parametrize("evaluation", ["mit", "prob"])
def test_one_process_route_equals_the_staged_route_on_the_host_stand_ins(tmp_path, stand_in_bin, evaluation):
    (tmp_path / "scenario").mkdir()
    d, records, bed, tus, targets = build_scenario(tmp_path / "scenario")
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    lines, k = [], 0
    other = ["0|0", "1|1", "0|1", "1|0", "./.", "1/1"]
    for line in (d / "in.vcf").read_text().splitlines():
        if line.startswith("##"):
            lines.append(line)
        elif line.startswith("#CHROM"):
            lines.append(line + "\tS1")
        else:
            lines.append(line + "\t" + other[k % len(other)])
            k += 1
    vcf2 = tmp_path / "two.vcf"
    vcf2.write_text("\n".join(lines) + "\n")
    for case, extra in (("ref", []), ("one", ["-f", str(d / "in.vcf"), "-s", "0"]), ("two", ["-f", str(vcf2), "-s", "all"]), ("pam", ["-p", "AG"])):
        got = {}
        for route in ("inproc", "staged"):
            out = tmp_path / ("%s_%s.txt" % (case, route))
            cmd = ["bash", driver, "-b", str(d / "targets.bed"), "-o", str(out), "-g", str(d / "genome.fa"), "-i", str(tmp_path / "idx"),
                   "-m", "5", "-t", "2", "-T", str(tmp_path / ("tmp_" + route)), "-a", str(d / "activity.txt"), "-e", evaluation] + extra
            env = dict(os.environ, VARSCOT_BIN=str(stand_in_bin), VARSCOT_RF_MODEL=os.path.join(ROOT, "varscot_amd", "models", "rfClassifier.vscrf"))
            if route == "staged":
                env["VARSCOT_STAGED"] = "1"
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout + r.stderr
            stems = ["%s_sample%d" % (str(out)[:-4], k) for k in (0, 1)] if case == "two" else [str(out)[:-4]]
            files = {}
            for stem in stems:
                files[os.path.basename(stem).replace(route, "")] = open(stem + ".txt", "rb").read()
                if evaluation != "mit":
                    files[os.path.basename(stem).replace(route, "") + "_fm"] = open(stem + "_feature_matrix.txt", "rb").read()
            got[route] = files
        assert got["inproc"] == got["staged"], case
        first = next(iter(got["inproc"].values())).decode().splitlines()
        assert len(first) > 20
        if case in ("one", "two"):
            assert any(line.endswith("REF") for line in first[1:]) and any("VAR_" in line.split("\t")[-1] for line in first[1:])


@pytest.mark.parametrize("seed", [20240, 7, 99])
def test_sam_text_of_the_stand_in_mapping_equals_the_oracles(tmp_path, stand_in_bin, oracle, seed):
    """bidir_mapping's own code - reads FASTA, packed genome, records in vsc_sam_order's order, flags (primary / secondary), NM and
    MD of both styles, the text written on several threads - over the stand-in search: on these scenarios (no N runs, nothing at
    a contig end, where the brute-force stand-in and the reference's rule would part) the hits are the oracle's, so the SAM text
    must be the oracle's byte for byte.  (Not so on the SNP genome: its windows are short contigs, hits at their right end fall
    under bidir_mapping.cpp:51-52, which the stand-in does not know - the real search against the oracle there is
    tests/test_tools.py's and tests/test_gpu_parity.py's, on the GPU.)"""
    d, records, bed, tus, targets = build_scenario(tmp_path, seed)
    for fasta, recs in (("genome.fa", records),):
        prefix = str(d / fasta.replace(".fa", "_idx"))
        assert subprocess.run([os.path.join(BIN, "bidir_index"), "-G", str(d / fasta), "-I", prefix], capture_output=True).returncode == 0
        for style in (0, 1):
            r = subprocess.run([str(stand_in_bin / "bidir_mapping"), "-G", str(d / fasta), "-I", prefix, "-R", str(d / "targets.fa"), "-M", "5", "-T", "3",
                                "-O", str(d / "out.sam"), "--md-style", str(style)], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            want = oracle.search_sam([s for _, s in recs], [n for n, _ in recs], [t[4] for t in targets], [t[0] for t in targets], 5, None, style)
            assert (d / "out.sam").read_text() == want and len(want.splitlines()) > 10
