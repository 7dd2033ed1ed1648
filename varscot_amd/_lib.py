"""ctypes binding of libvarscot_hip.so (include/varscot_hip.h).

The library is the product: there is no Python or CPU implementation of the search behind this
module.  Importing it fails loudly when the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C varscot_amd/csrc`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSC_LIB_PATH") or os.path.join(_HERE, "libvarscot_hip.so")  # override for A/B experiments

VSC_OK = 0
ERRORS = {-22: "VSC_ERR_INVALID", -12: "VSC_ERR_NOMEM", -5: "VSC_ERR_DEVICE", -34: "VSC_ERR_RANGE",
          -19: "VSC_ERR_NODEVICE"}

HIT_DTYPE = np.dtype([("guide", "<u4"), ("contig", "<u4"), ("pos", "<u4"), ("info", "<u4")])
CONTIG_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u4"), ("reserved", "<u4")])
N_FEATURES = 442


class SearchParams(C.Structure):
    _fields_ = [("max_mismatches", C.c_uint32), ("has_extra_pam", C.c_uint8), ("extra_pam", C.c_char * 2),
                ("algorithm", C.c_uint8)]


ALGO_AUTO, ALGO_SCAN, ALGO_SEED = 0, 1, 2


class RfModel(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("n_nodes", C.c_uint32), ("node_status", C.c_void_p), ("feature", C.c_void_p),
                ("left", C.c_void_p), ("right", C.c_void_p), ("split", C.c_void_p), ("node_class", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("prep_ms", C.c_double), ("sort_ms", C.c_double),
                ("finalize_ms", C.c_double), ("score_ms", C.c_double), ("total_ms", C.c_double),
                ("index_ms", C.c_double), ("sites", C.c_uint64), ("pairs", C.c_uint64), ("hits", C.c_uint64),
                ("genome_bytes", C.c_uint64), ("passes", C.c_uint32), ("algorithm", C.c_uint32),
                ("sort_bytes", C.c_uint64), ("sort_levels", C.c_uint32), ("sort_bin_bits", C.c_uint32),
                ("read_passes", C.c_uint32), ("sort_fallbacks", C.c_uint32), ("list_entries", C.c_uint64), ("seed_cut", C.c_uint32),
                ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class MultiTiming(C.Structure):
    _fields_ = [("search_wall_ms", C.c_double), ("search_ms_max", C.c_double), ("exchange_ms", C.c_double),
                ("merge_ms", C.c_double), ("total_ms", C.c_double), ("hits", C.c_uint64), ("exchanged_bytes", C.c_uint64),
                ("n_devices", C.c_uint32), ("used_rccl", C.c_uint32), ("score_ms_max", C.c_double), ("callback_ms", C.c_double),
                ("batches", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class MultiScore(C.Structure):
    """vsc_multi_score: what every shard computes per hit before the exchange (vsc_multi_search_stream)."""
    _fields_ = [("mode", C.c_uint32), ("reserved", C.c_uint32), ("guide_activity", C.c_void_p), ("model", C.POINTER(RfModel))]


MULTI_SCORE_NONE, MULTI_SCORE_ROWS, MULTI_SCORE_VOTES = 0, 1, 2


class DebugParams(C.Structure):
    """vsc_debug_params (include/varscot_hip_debug.h): test / experiment hooks; 0 or -1 = the library's default."""
    _fields_ = [("seed_groups_per_cu", C.c_uint32), ("seed_reserve", C.c_uint32), ("sort_cap", C.c_uint32),
                ("sort_max_bits", C.c_uint32), ("sort_xcd", C.c_int32), ("sort_debug", C.c_uint32),
                ("sort_optimistic", C.c_int32), ("sort_slot_cap", C.c_uint32), ("score_chunk", C.c_uint64),
                ("score_slices", C.c_int32), ("score_slice_shift", C.c_uint32), ("seed_shared", C.c_int32),
                ("seed_group_out", C.c_int32), ("seed_tight", C.c_int32), ("rf_form", C.c_int32),
                ("reserved", C.c_uint32 * 1)]
    SIGNED_DEFAULT = ("sort_xcd", "sort_optimistic", "score_slices", "seed_shared", "seed_group_out", "seed_tight", "rf_form")

    @classmethod
    def defaults(cls):
        d = cls()
        for k in cls.SIGNED_DEFAULT:
            setattr(d, k, -1)
        return d


class MultiDebugParams(C.Structure):
    _fields_ = [("rccl", C.c_int32), ("rccl_library", C.c_char_p)]


BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32)  # vsc_batch_fn
ROWS_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)  # vsc_rows_batch_fn
MULTI_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)  # vsc_multi_batch_fn

# every symbol include/varscot_hip.h declares: (name, restype, argtypes)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p
SYMBOLS = [
    ("vsc_abi_version", C.c_int, []),
    ("vsc_device_count", C.c_int, []),
    ("vsc_ctx_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("vsc_ctx_destroy", C.c_int, [_vp]),
    ("vsc_ctx_release_scratch", C.c_int, [_vp]),
    ("vsc_ctx_set_stream", C.c_int, [_vp, _vp]),
    ("vsc_last_error", C.c_char_p, [_vp]),
    ("vsc_ctx_timing", C.c_int, [_vp, C.POINTER(Timing)]),
    ("vsc_layout_contigs", C.c_uint64, [_vp, C.c_uint32, _vp]),
    ("vsc_planes_init", None, [_vp, _vp, _vp, C.c_uint64]),
    ("vsc_pack_bases", None, [C.c_char_p, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_unpack_bases", None, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    ("vsc_pack_guide", C.c_uint64, [C.c_char_p]),
    ("vsc_genome_load", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_uint32,
                                  C.POINTER(_vp)]),
    ("vsc_genome_free", C.c_int, [_vp]),
    ("vsc_genome_build_index", C.c_int, [_vp, _vp, C.POINTER(SearchParams)]),
    ("vsc_genome_index_save", C.c_int, [_vp, _vp, C.c_char_p]),
    ("vsc_genome_index_load", C.c_int, [_vp, _vp, C.c_char_p]),
    ("vsc_genome_device_bytes", C.c_uint64, [_vp]),
    ("vsc_search", C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(SearchParams), C.POINTER(_vp)]),
    ("vsc_search_stream", C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(SearchParams), C.c_uint32, BATCH_FN, _vp]),
    ("vsc_search_stream_rows", C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(SearchParams), C.c_uint32, ROWS_BATCH_FN, _vp]),
    ("vsc_hits_count", C.c_uint64, [_vp]),
    ("vsc_hits_data_dev", _vp, [_vp]),
    ("vsc_hits_data", C.c_int, [_vp, C.POINTER(_vp)]),
    ("vsc_hits_copy", C.c_int, [_vp, _vp, C.c_int]),
    ("vsc_hits_merge", C.c_int, [_vp, _vp, C.c_int, _vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    ("vsc_hits_pack_exchange", C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.c_int, _vp]),
    ("vsc_hits_merge_packed", C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    ("vsc_hits_merge_packed_votes", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp), _vp, C.c_int]),
    ("vsc_hits_free", C.c_int, [_vp]),
    ("vsc_score_hits", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_score_hits_packed", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_unpack_features", None, [_vp, C.c_uint64, _vp]),
    ("vsc_score_pairs", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    ("vsc_multi_destroy", C.c_int, [_vp]),
    ("vsc_multi_release_scratch", C.c_int, [_vp]),
    ("vsc_multi_size", C.c_int, [_vp]),
    ("vsc_multi_ctx", _vp, [_vp, C.c_int]),
    ("vsc_multi_result_ctx", _vp, [_vp]),
    ("vsc_multi_last_error", C.c_char_p, [_vp]),
    ("vsc_multi_uses_rccl", C.c_int, [_vp]),
    ("vsc_multi_get_timing", C.c_int, [_vp, C.POINTER(MultiTiming)]),
    ("vsc_multi_genome_load", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32, C.POINTER(_vp)]),
    ("vsc_multi_genome_free", C.c_int, [_vp]),
    ("vsc_multi_genome_build_index", C.c_int, [_vp, _vp, C.POINTER(SearchParams)]),
    ("vsc_multi_search", C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(SearchParams), C.POINTER(_vp)]),
    ("vsc_multi_search_stream", C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(SearchParams), C.c_uint32, C.POINTER(MultiScore), MULTI_BATCH_FN, _vp]),
    ("vsc_windows_build", C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, C.c_uint32,
                                    C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    ("vsc_windows_count", C.c_uint32, [_vp]),
    ("vsc_windows_words", C.c_uint64, [_vp]),
    ("vsc_windows_plane", _vp, [_vp, C.c_int]),
    ("vsc_windows_contigs", _vp, [_vp]),
    ("vsc_windows_name", _vp, [_vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("vsc_windows_name_offsets", _vp, [_vp]),
    ("vsc_windows_free", None, [_vp]),
    ("vsc_rf_predict", C.c_int, [_vp, C.POINTER(RfModel), _vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_rf_predict_packed", C.c_int, [_vp, C.POINTER(RfModel), _vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_score_classify_hits", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _vp, C.POINTER(RfModel), C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    ("vsc_sam_order", None, [_vp, C.c_uint64, _vp, _vp]),
]
# include/varscot_hip_debug.h (test / experiment hooks, not part of the drop-in boundary)
DEBUG_SYMBOLS = [
    ("vsc_ctx_set_debug_params", C.c_int, [_vp, C.POINTER(DebugParams)]),
    ("vsc_ctx_get_debug_params", C.c_int, [_vp, C.POINTER(DebugParams)]),
    ("vsc_debug_set_host_timing", None, [C.c_int]),
    ("vsc_ctx_create_masked", C.c_int, [C.c_int, _vp, C.c_uint32, C.POINTER(_vp)]),
    ("vsc_multi_create_debug", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(MultiDebugParams), C.POINTER(_vp)]),
]

_lib = None


class VarscotError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__("%s (%d)%s" % (ERRORS.get(code, "VSC_ERR"), code, ": " + message if message else ""))


def lib():
    """The loaded shared library.  Raises ImportError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libvarscot_hip.so is not built (%s). Build it with `make -C varscot_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own copy of the HIP runtime.  Two runtimes in one process do not share
        # the device: whichever initialises it second reports "no GPU".  With torch imported first the
        # dynamic loader resolves this library's libamdhip64 dependency to the copy torch already
        # loaded, so there is ONE runtime whatever the later order of initialisation.  (varscot_amd.dist
        # and bench.py need torch anyway; the C++ tools never load it.)
        if os.environ.get("VSC_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS + DEBUG_SYMBOLS:
            fn = getattr(L, name)  # AttributeError = header and library out of sync
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(code, ctx=None):
    if code != VSC_OK:
        msg = ""
        if ctx:
            msg = lib().vsc_last_error(ctx).decode(errors="replace")
        raise VarscotError(code, msg)


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
