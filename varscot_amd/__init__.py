"""varscot_amd - MI355X-native implementation of VARSCOT's genome-wide off-target search hot path.

The compute lives in libvarscot_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/varscot_hip.h); this package is the thin host-side mirror used by tests, bench.py and the
Python entry points.  There is no CPU fallback: without the built library importing fails.
"""
from ._lib import HIT_DTYPE, CONTIG_DTYPE, N_FEATURES, LIB_PATH, VarscotError, lib  # noqa: F401
from .api import (Context, Genome, Hits, MultiContext, PackedGenome, device_count, pack_guides, sam_order,  # noqa: F401
                  unpack_features, variant_windows)

lib()  # fail loudly at import time if the HIP extension is missing
