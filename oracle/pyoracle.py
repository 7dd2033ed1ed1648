"""ctypes binding of oracle/libvsc_oracle.so (the CPU restatement of the reference's hot path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under varscot_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvsc_oracle.so")

HIT_DTYPE = np.dtype([("guide", "<u4"), ("contig", "<u4"), ("pos", "<u4"), ("info", "<u4")])

MODE_PREDICATE = 0
MODE_REFERENCE_FLOW = 1


def build():
    """(Re)build the oracle with gcc; returns the path of the shared library."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libvsc_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        pp = C.POINTER(C.c_char_p)
        u32p = C.POINTER(C.c_uint32)
        L.orc_search.restype = C.c_long
        L.orc_search.argtypes = [pp, u32p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p,
                                 C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.orc_search_fast.restype = C.c_long
        L.orc_search_fast.argtypes = [pp, u32p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p,
                                      C.c_int, C.c_void_p, C.c_long]
        L.orc_count_fast.restype = C.c_long
        L.orc_count_fast.argtypes = [pp, u32p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p,
                                     C.c_int, C.POINTER(C.c_long)]
        L.orc_search_sam.restype = C.c_void_p
        L.orc_search_sam.argtypes = [pp, u32p, pp, C.c_uint32, C.c_char_p, pp, C.c_uint32, C.c_uint32,
                                     C.c_char_p, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_md_string.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
        L.orc_md_positions.restype = C.c_int
        L.orc_md_positions.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.orc_mit_score.restype = C.c_double
        L.orc_mit_score.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.orc_feature_row.argtypes = [C.c_char_p, C.c_char_p, u32p]
        L.orc_max_threads.restype = C.c_int
        L.orc_planes_to_text.restype = None
        L.orc_planes_to_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_pigeon_build.restype = C.c_void_p
        L.orc_pigeon_build.argtypes = [pp, u32p, C.c_uint32]
        L.orc_pigeon_free.argtypes = [C.c_void_p]
        L.orc_pigeon_search.restype = C.c_long
        L.orc_pigeon_search.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_void_p,
                                        C.c_long, C.POINTER(C.c_long)]
        _lib = L
    return _lib


def _genome_args(contigs):
    bufs = [c if isinstance(c, (bytes, np.ndarray)) else c.encode() for c in contigs]
    # numpy uint8 arrays (planes_to_text of a whole chromosome) are passed by address, not copied
    arr = (C.c_char_p * max(1, len(bufs)))(*[C.cast(b.ctypes.data, C.c_char_p) if isinstance(b, np.ndarray) else b for b in bufs])
    lens = (C.c_uint32 * max(1, len(bufs)))(*[len(b) for b in bufs])
    return bufs, arr, lens


def _guides_arg(guides):
    gs = [g if isinstance(g, bytes) else g.encode() for g in guides]
    assert all(len(g) == 23 for g in gs), "guides must be 23 nt"
    return b"".join(gs)


def _pam(extra_pam):
    if not extra_pam:
        return None
    return extra_pam if isinstance(extra_pam, bytes) else extra_pam.encode()


def search(contigs, guides, max_mm, extra_pam=None, mode=MODE_PREDICATE, compat_u16=False):
    """Character-level restatement (slow).  Returns a structured array of hits."""
    bufs, arr, lens = _genome_args(contigs)
    g = _guides_arg(guides)
    L = lib()
    n = L.orc_search(arr, lens, len(bufs), g, len(guides), max_mm, _pam(extra_pam), mode, int(compat_u16), None, 0)
    if n < 0:
        raise ValueError("orc_search failed (max_mm outside 0..8?)")
    out = np.zeros(n, dtype=HIT_DTYPE)
    if n:
        L.orc_search(arr, lens, len(bufs), g, len(guides), max_mm, _pam(extra_pam), mode, int(compat_u16),
                     out.ctypes.data_as(C.c_void_p), n)
    return out


def search_fast(contigs, guides, max_mm, extra_pam=None, threads=0, cap=None):
    """Bit-parallel OpenMP port; hits ascending by (guide, strand, contig, pos)."""
    bufs, arr, lens = _genome_args(contigs)
    g = _guides_arg(guides)
    L = lib()
    if cap is None:
        sites = C.c_long(0)
        cap = L.orc_count_fast(arr, lens, len(bufs), g, len(guides), max_mm, _pam(extra_pam), threads,
                               C.byref(sites))
    out = np.zeros(max(cap, 0), dtype=HIT_DTYPE)
    n = L.orc_search_fast(arr, lens, len(bufs), g, len(guides), max_mm, _pam(extra_pam), threads,
                          out.ctypes.data_as(C.c_void_p), len(out))
    if n < 0:
        raise ValueError("orc_search_fast failed")
    return out[:min(n, len(out))]


def planes_to_text(hi, lo, nmask, pos, n):
    """n characters from global position pos of packed planes (uint32 arrays; layout of include/varscot_hip.h,
    restated in vsc_planes.c independently of the product's vsc_unpack_bases) as a bytes-like numpy uint8 array."""
    hi, lo, nmask = (np.ascontiguousarray(a, dtype=np.uint32) for a in (hi, lo, nmask))
    assert pos + n <= 32 * len(hi)
    out = np.empty(n, dtype=np.uint8)
    lib().orc_planes_to_text(hi.ctypes.data, lo.ctypes.data, nmask.ctypes.data, pos, n, out.ctypes.data)
    return out


def count_fast(contigs, guides, max_mm, extra_pam=None, threads=0):
    """(hits, sites) without storing the hits - used by the cpu_baseline leg of bench.py."""
    bufs, arr, lens = _genome_args(contigs)
    g = _guides_arg(guides)
    sites = C.c_long(0)
    n = lib().orc_count_fast(arr, lens, len(bufs), g, len(guides), max_mm, _pam(extra_pam), threads,
                             C.byref(sites))
    return n, sites.value


class PigeonIndex:
    """The k-mer tables of vsc_pigeon.c over a text (built once, searched many times)."""

    def __init__(self, contigs):
        self._bufs, arr, lens = _genome_args(contigs)
        self._h = lib().orc_pigeon_build(arr, lens, len(self._bufs))
        if not self._h:
            raise MemoryError("orc_pigeon_build failed")

    def search(self, guides, max_mm, extra_pam=None, threads=0):
        """Hits sorted by (guide, strand, contig, pos), and the number of delegate calls."""
        g = _guides_arg(guides)
        cand = C.c_long(0)
        n = lib().orc_pigeon_search(self._h, g, len(guides), max_mm, _pam(extra_pam), threads, None, 0, C.byref(cand))
        if n < 0:
            raise ValueError("orc_pigeon_search failed")
        out = np.zeros(n, dtype=HIT_DTYPE)
        if n:
            lib().orc_pigeon_search(self._h, g, len(guides), max_mm, _pam(extra_pam), threads,
                                    out.ctypes.data_as(C.c_void_p), n, C.byref(cand))
        return out, cand.value

    def count(self, guides, max_mm, extra_pam=None, threads=0):
        g = _guides_arg(guides)
        cand = C.c_long(0)
        n = lib().orc_pigeon_search(self._h, g, len(guides), max_mm, _pam(extra_pam), threads, None, 0, C.byref(cand))
        return n, cand.value

    def close(self):
        if self._h:
            lib().orc_pigeon_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_sam(contigs, contig_names, guides, guide_names, max_mm, extra_pam=None, md_style=0):
    bufs, arr, lens = _genome_args(contigs)
    g = _guides_arg(guides)
    cn = (C.c_char_p * max(1, len(bufs)))(*[n.encode() for n in contig_names])
    gn = (C.c_char_p * max(1, len(guides)))(*[n.encode() for n in guide_names])
    L = lib()
    p = L.orc_search_sam(arr, lens, cn, len(bufs), g, gn, len(guides), max_mm, _pam(extra_pam), md_style)
    if not p:
        raise ValueError("orc_search_sam failed")
    s = C.string_at(p).decode()
    L.orc_free(p)
    return s


def md_string(window, read, md_style=0):
    out = C.create_string_buffer(64)
    lib().orc_md_string(window.encode(), read.encode(), md_style, out)
    return out.value.decode()


def md_positions(md):
    out = (C.c_int * 24)()
    n = lib().orc_md_positions(md.encode(), out)
    return [out[i] for i in range(min(n, 24))]


def mit_score(positions):
    """Returns (score, reference_ub_flag)."""
    arr = (C.c_int * max(1, len(positions)))(*positions)
    ub = C.c_int(0)
    s = lib().orc_mit_score(arr, len(positions), C.byref(ub))
    return s, bool(ub.value)


def feature_row(on, off):
    out = np.zeros(442, dtype=np.uint32)
    lib().orc_feature_row(on.encode(), off.encode(), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def max_threads():
    return lib().orc_max_threads()


def sort_key(h):
    """Ascending (guide, strand, contig, pos) order of a hit array."""
    return np.lexsort((h["pos"], h["contig"], h["info"] >> 31, h["guide"]))
