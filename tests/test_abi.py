"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares, and its
host-side helpers (packing, SAM ordering) agree with the oracle.  No compute entry point is called."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import varscot_amd as va
from varscot_amd import _lib
from helpers import make_genome, random_guides, random_seq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return text, set(re.findall(r"\b(vsc_[a-z0-9_]+)\s*\(", text))


def test_header_symbols_are_exported_and_bound():
    """Every include/*.h: what it declares is exported by the library and bound by the Python mirror."""
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["varscot_hip.h", "varscot_hip_debug.h"]
    text, declared = _declared("varscot_hip.h")
    assert declared == {name for name, _, _ in _lib.SYMBOLS}
    _, declared_dbg = _declared("varscot_hip_debug.h")
    assert declared_dbg == {name for name, _, _ in _lib.DEBUG_SYMBOLS}
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared | declared_dbg:
        assert hasattr(L, name), name
    version = int(re.search(r"#define\s+VSC_ABI_VERSION\s+(\d+)", text).group(1))
    assert va.lib().vsc_abi_version() == version == 5


def test_the_library_reads_no_environment_variable():
    """Test hooks are explicit calls (varscot_hip_debug.h): a stray VSC_* variable in a user's environment must
    not steer kernels or buffer sizes."""
    csrc = os.path.join(ROOT, "varscot_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".cpp", ".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, name)).read(), name


def test_graft_entry_build_check_follows_the_header():
    """__graft_entry__.build() must not pin a version number of its own (it did: 1, while the header said 2)."""
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "VSC_ABI_VERSION" in src and not re.search(r"vsc_abi_version\(\)\s*==\s*\d", src)


def test_no_device_fails_loudly():
    if va.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(va.VarscotError) as e:
        va.Context(0)
    assert e.value.code == -19


def test_pack_roundtrip_and_layout():
    rng = np.random.default_rng(5)
    seqs = [random_seq(rng, n) for n in (1, 31, 32, 33, 64, 1000, 23)]
    seqs[5] = seqs[5][:100] + "NNNNnnnn" + seqs[5][108:500] + "acgtRYK" + seqs[5][507:]
    g = va.PackedGenome.from_sequences(seqs)
    off = 0
    for c, s in enumerate(seqs):
        assert int(g.contigs["offset"][c]) == off and int(g.contigs["length"][c]) == len(s)
        want = "".join(ch.upper() if ch.upper() in "ACGT" else "N" for ch in s)
        assert g.contig_sequence(c) == want
        assert g.decode(off + len(s), 1) == "N"  # separator
        off += len(s) + 1
    assert g.n_words == (off + 31) // 32


def test_pack_guide_codes():
    code = int(va.pack_guides(["ACGTNACGTACGTACGTACGTGG"])[0])
    want = "ACGTAACGTACGTACGTACGTGG"  # N -> A
    assert [(code >> (2 * i)) & 3 for i in range(23)] == ["ACGT".index(c) for c in want]
    with pytest.raises(ValueError):
        va.pack_guides(["ACGT"])


def test_shard_words_partition_the_planes():
    g = va.PackedGenome.from_sequences(["A" * 100000])
    for world in (1, 2, 3, 8):
        ranges = [g.shard_words(r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == g.n_words
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and b % 64 == 0


@pytest.mark.parametrize("seed,max_mm", [(3, 4), (4, 8)])
def test_sam_order_matches_reference_flow(oracle, seed, max_mm):
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 5)
    contigs = make_genome(seed, [3000, 1500, 40], guides, max_mm, n_plant=80)
    asc = oracle.search(contigs, guides, max_mm, mode=oracle.MODE_PREDICATE)
    flow = oracle.search(contigs, guides, max_mm, mode=oracle.MODE_REFERENCE_FLOW)
    order, sec = va.sam_order(asc)
    got = asc[order.astype(np.int64)]
    for f in ("guide", "contig", "pos"):
        assert np.array_equal(got[f], flow[f])
    assert np.array_equal(got["info"] | (sec.astype(np.uint32) << 30), flow["info"])
