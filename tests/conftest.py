import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The built artefacts are git-ignored; build them in-tree if a fresh checkout lacks them."""
    need = [os.path.join(ROOT, "varscot_amd", "libvarscot_hip.so"),
            os.path.join(ROOT, "varscot_amd", "bin", "bidir_index"),
            os.path.join(ROOT, "varscot_amd", "bin", "bidir_mapping"),
            os.path.join(ROOT, "varscot_amd", "bin", "vcf_loader"),
            os.path.join(ROOT, "varscot_amd", "bin", "bam_merger"),
            os.path.join(ROOT, "varscot_amd", "bin", "fasta_writer"),
            os.path.join(ROOT, "varscot_amd", "bin", "classification_pipeline"),
            os.path.join(ROOT, "varscot_amd", "bin", "bam_merger_ref_only"),
            os.path.join(ROOT, "oracle", "libvsc_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/), built on demand.  Checker only - never the product path."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
