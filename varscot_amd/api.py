"""Host-side Python mirror of the C ABI (include/varscot_hip.h): packed genome, device context,
search and per-hit scoring.  Thin plumbing only - every computation happens in libvarscot_hip.so.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import (ALGO_AUTO, ALGO_SCAN, ALGO_SEED, HIT_DTYPE, CONTIG_DTYPE, N_FEATURES, SearchParams, Timing, check,
                   lib, ptr)

READ_LEN = 23
TILE_WORDS = 64  # shard boundaries are multiples of this many 32-base words (one scan tile)


def pack_guides(guides):
    """23-nt reads -> uint64 codes (vsc_pack_guide; non-ACGT letters become A like SeqAn's Dna)."""
    L = lib()
    out = np.empty(len(guides), dtype=np.uint64)
    for i, g in enumerate(guides):
        b = g if isinstance(g, bytes) else g.encode()
        if len(b) != READ_LEN:
            raise ValueError("read %d has length %d; VARSCOT searches 23-nt reads (20 nt + PAM)" % (i, len(b)))
        out[i] = L.vsc_pack_guide(b)
    return out


class PackedGenome:
    """The host-side packed planes of a genome (0.375 byte per base) plus its contig table.

    Counterpart of the reference's on-disk index (read_mapping/bidir_index.cpp) - here three bit
    planes (hi, lo, nmask) over one global coordinate space, see include/varscot_hip.h.
    """

    def __init__(self, hi, lo, nmask, contigs, names=None):
        self.hi, self.lo, self.nmask = hi, lo, nmask
        self.contigs = np.ascontiguousarray(contigs, dtype=CONTIG_DTYPE)
        self.names = list(names) if names is not None else ["contig%d" % i for i in range(len(self.contigs))]

    @property
    def n_words(self):
        return len(self.hi)

    @property
    def n_bases(self):
        return int(self.contigs["length"].sum())

    @classmethod
    def from_sequences(cls, seqs, names=None):
        L = lib()
        bufs = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        lens = np.array([len(b) for b in bufs], dtype=np.uint32)
        table = np.zeros(len(bufs), dtype=CONTIG_DTYPE)
        n_words = int(L.vsc_layout_contigs(ptr(lens), len(bufs), ptr(table)))
        n_words = max(n_words, 1)
        hi = np.empty(n_words, dtype=np.uint32)
        lo = np.empty(n_words, dtype=np.uint32)
        nm = np.empty(n_words, dtype=np.uint32)
        L.vsc_planes_init(ptr(hi), ptr(lo), ptr(nm), n_words)
        for b, row in zip(bufs, table):
            L.vsc_pack_bases(b, len(b), int(row["offset"]), ptr(hi), ptr(lo), ptr(nm))
        return cls(hi, lo, nm, table, names)

    @classmethod
    def from_index_file(cls, prefix):
        """Reads <prefix>.vsc as written by the bidir_index tool (tools/vsc_host.hpp)."""
        with open(prefix + ".vsc", "rb") as f:
            magic = f.read(8)
            assert magic in (b"VSCIDX01", b"VSCIDX02"), "not a packed genome"
            nc, nw = (int(x) for x in np.frombuffer(f.read(16), dtype="<u8"))
            if magic == b"VSCIDX02":
                f.read(16)  # size and modification time of the FASTA it was packed from
            table = np.frombuffer(f.read(nc * CONTIG_DTYPE.itemsize), dtype=CONTIG_DTYPE).copy()
            names, name_bytes = [], 0
            for _ in range(nc):
                ln = int(np.frombuffer(f.read(4), dtype="<u4")[0])
                names.append(f.read(ln).decode())
                name_bytes += 4 + ln
            if magic == b"VSCIDX02":
                f.read((8 - name_bytes % 8) % 8)  # the planes start 8-byte aligned
            hi = np.frombuffer(f.read(nw * 4), dtype="<u4").copy()
            lo = np.frombuffer(f.read(nw * 4), dtype="<u4").copy()
            nm = np.frombuffer(f.read(nw * 4), dtype="<u4").copy()
        return cls(hi, lo, nm, table, names)

    def decode(self, pos, n):
        """n characters starting at global position pos (N outside contigs)."""
        out = C.create_string_buffer(n)
        lib().vsc_unpack_bases(ptr(self.hi), ptr(self.lo), ptr(self.nmask), pos, n, out)
        return out.raw.decode()

    def contig_sequence(self, c):
        row = self.contigs[c]
        return self.decode(int(row["offset"]), int(row["length"]))

    def shard_words(self, rank, world):
        """Tile-aligned word range [begin, end) of the planes that rank `rank` of `world` owns."""
        tiles = (self.n_words + TILE_WORDS - 1) // TILE_WORDS
        b = (tiles * rank // world) * TILE_WORDS
        e = (tiles * (rank + 1) // world) * TILE_WORDS
        return min(b, self.n_words), min(e, self.n_words)


class Context:
    """One device + one stream (vsc_ctx).  Single-threaded."""

    def __init__(self, device=0, cu_mask=None):
        """cu_mask (experiments, include/varscot_hip_debug.h): uint32 words, bit i of word i // 32 = compute unit i may be used."""
        self._h = C.c_void_p()
        if cu_mask is None:
            check(lib().vsc_ctx_create(device, C.byref(self._h)))
        else:
            mask = np.ascontiguousarray(cu_mask, dtype=np.uint32)
            check(lib().vsc_ctx_create_masked(device, mask.ctypes.data, len(mask), C.byref(self._h)))
        self.device = device
        self._children = weakref.WeakSet()  # genomes and results must go before their context

    def close(self):
        if self._h:
            for child in sorted(self._children, key=lambda c: isinstance(c, Genome)):
                child.close()  # results first, then genomes
            lib().vsc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_handle):
        check(lib().vsc_ctx_set_stream(self._h, C.c_void_p(hip_stream_handle)), self._h)

    def timing(self):
        t = Timing()
        check(lib().vsc_ctx_timing(self._h, C.byref(t)), self._h)
        return t.as_dict()

    def release_scratch(self):
        """vsc_ctx_release_scratch: give the pooled scratch and record buffers back to the device."""
        check(lib().vsc_ctx_release_scratch(self._h), self._h)

    def set_debug(self, **hooks):
        """Test / experiment hooks of include/varscot_hip_debug.h (vsc_ctx_set_debug_params): e.g.
        set_debug(sort_cap=16) forces many sort levels on a few thousand records.  No arguments: back to the
        defaults.  The library reads no environment variable."""
        d = _lib.DebugParams.defaults()
        for k, v in hooks.items():
            if k not in dict(_lib.DebugParams._fields_) or k == "reserved":
                raise TypeError("unknown debug hook %r" % k)
            setattr(d, k, int(v))
        check(lib().vsc_ctx_set_debug_params(self._h, C.byref(d) if hooks else None), self._h)

    def load_genome(self, packed, rank=0, world=1):
        return Genome(self, packed, rank, world)

    def score_pairs(self, on_targets, off_targets, masks, mit=True, features=False):
        """vsc_score_pairs: MIT score / feature rows of explicit (on-target, off-target) 23-mer pairs."""
        on = np.ascontiguousarray(on_targets if isinstance(on_targets, np.ndarray) else pack_guides(on_targets), dtype=np.uint64)
        off = np.ascontiguousarray(off_targets if isinstance(off_targets, np.ndarray) else pack_guides(off_targets), dtype=np.uint64)
        mk = np.ascontiguousarray(masks, dtype=np.uint32)
        n = len(on)
        assert len(off) == n and len(mk) == n
        m = np.empty(n, dtype=np.float64) if mit else None
        fl = np.empty(n, dtype=np.uint8) if mit else None
        ft = np.empty((n, N_FEATURES), dtype=np.uint8) if features else None
        check(lib().vsc_score_pairs(self._h, ptr(on), ptr(off), ptr(mk), n, ptr(m), ptr(fl), ptr(ft)), self._h)
        return m, fl, ft


class Genome:
    """(A shard of) a packed genome resident in HBM (vsc_genome)."""

    def __init__(self, ctx, packed, rank=0, world=1):
        b, e = packed.shard_words(rank, world)
        if e <= b:
            raise ValueError("rank %d of %d owns no words of this genome" % (rank, world))
        halo_end = min(e + 1, packed.n_words)  # 22-base halo = 1 word
        self._load(ctx, packed.hi[b:halo_end], packed.lo[b:halo_end], packed.nmask[b:halo_end], b, e - b,
                   packed.contigs)
        self.packed = packed

    @classmethod
    def from_shard(cls, ctx, hi, lo, nmask, first_word, own_words, contigs):
        """Planes of words [first_word, first_word + len(hi)) of which the first own_words are owned."""
        self = cls.__new__(cls)
        self.packed = None
        self._load(ctx, hi, lo, nmask, first_word, own_words, contigs)
        return self

    def _load(self, ctx, hi, lo, nm, first_word, own_words, contigs):
        self.ctx = ctx
        self.first_word, self.own_words = first_word, own_words
        hi = np.ascontiguousarray(hi, dtype=np.uint32)
        lo = np.ascontiguousarray(lo, dtype=np.uint32)
        nm = np.ascontiguousarray(nm, dtype=np.uint32)
        contigs = np.ascontiguousarray(contigs, dtype=CONTIG_DTYPE)
        self._h = C.c_void_p()
        check(lib().vsc_genome_load(ctx._h, ptr(hi), ptr(lo), ptr(nm), first_word, len(hi), own_words, ptr(contigs),
                                    len(contigs), C.byref(self._h)), ctx._h)
        ctx._children.add(self)

    @property
    def device_bytes(self):
        return int(lib().vsc_genome_device_bytes(self._h))

    def close(self):
        if self._h:
            lib().vsc_genome_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _params(max_mismatches, extra_pam, algorithm):
        p = SearchParams()
        p.max_mismatches = max_mismatches
        p.algorithm = {"auto": ALGO_AUTO, "scan": ALGO_SCAN, "seed": ALGO_SEED}.get(algorithm, algorithm)
        if extra_pam:
            e = extra_pam if isinstance(extra_pam, bytes) else extra_pam.encode()
            if len(e) != 2:
                raise ValueError("the additional PAM (-P) must be 2 letters")
            p.has_extra_pam = 1
            p.extra_pam = e
        return p

    def build_index(self, extra_pam=None):
        """Build the seed index now (vsc_genome_build_index); searches build it on demand otherwise."""
        p = self._params(0, extra_pam, ALGO_SEED)
        check(lib().vsc_genome_build_index(self.ctx._h, self._h, C.byref(p)), self.ctx._h)

    def save_index(self, path):
        """Write the resident seed index to a file (vsc_genome_index_save)."""
        check(lib().vsc_genome_index_save(self.ctx._h, self._h, os.fsencode(path)), self.ctx._h)

    def load_index(self, path):
        """Replace the seed index by a file's (vsc_genome_index_load); refused if it belongs to another genome."""
        check(lib().vsc_genome_index_load(self.ctx._h, self._h, os.fsencode(path)), self.ctx._h)

    def search(self, guides, max_mismatches, extra_pam=None, algorithm="auto"):
        """guides: list of 23-nt strings or a uint64 array from pack_guides().
        algorithm: "auto" | "scan" | "seed" - same records either way (see include/varscot_hip.h)."""
        codes = guides if isinstance(guides, np.ndarray) else pack_guides(guides)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        p = self._params(max_mismatches, extra_pam, algorithm)
        h = C.c_void_p()
        check(lib().vsc_search(self.ctx._h, self._h, ptr(codes), len(codes), C.byref(p), C.byref(h)), self.ctx._h)
        return Hits(self, h, codes)

    def search_streamed(self, guides, max_mismatches, on_batch, batch=0, extra_pam=None, algorithm="auto"):
        """vsc_search_stream: the reads are searched in batches of `batch` (0 = the library's maximum, 16 384)
        and on_batch(hits, first_guide, n_guides) is called with every batch's result - a Hits object that is
        only valid inside the call (score it, copy it out, gather it; the library frees it afterwards).
        Record order and read indices are those of one big search.  ctx.timing() afterwards holds sums."""
        codes = guides if isinstance(guides, np.ndarray) else pack_guides(guides)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        p = self._params(max_mismatches, extra_pam, algorithm)
        failure = []

        def trampoline(_user, handle, first, count):
            try:
                h = Hits(self, C.c_void_p(handle), codes, owned=False)
                try:
                    on_batch(h, int(first), int(count))
                finally:
                    h.close()
                return 0
            except BaseException as e:  # no exception may cross the C boundary
                failure.append(e)
                return -5

        cb = _lib.BATCH_FN(trampoline)
        rc = lib().vsc_search_stream(self.ctx._h, self._h, ptr(codes), len(codes), C.byref(p), int(batch), cb, None)
        if failure:
            raise failure[0]
        check(rc, self.ctx._h)


    def search_streamed_rows(self, guides, max_mismatches, on_batch, batch=0, extra_pam=None, algorithm="auto"):
        """vsc_search_stream_rows: as search_streamed, and every batch arrives with its 64-byte packed feature rows -
        on_batch(hits, first_guide, n_guides, rows_dev) with rows_dev = device address of len(hits) * 64 bytes (row i belongs
        to record i; valid inside the call; None for an empty batch).  The rows are written by the kernel that assembles the
        records - no second pass over the hits, no gather from the planes."""
        codes = guides if isinstance(guides, np.ndarray) else pack_guides(guides)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        p = self._params(max_mismatches, extra_pam, algorithm)
        failure = []

        def trampoline(_user, handle, first, count, rows_dev):
            try:
                h = Hits(self, C.c_void_p(handle), codes, owned=False)
                try:
                    on_batch(h, int(first), int(count), rows_dev)
                finally:
                    h.close()
                return 0
            except BaseException as e:  # no exception may cross the C boundary
                failure.append(e)
                return -5

        cb = _lib.ROWS_BATCH_FN(trampoline)
        rc = lib().vsc_search_stream_rows(self.ctx._h, self._h, ptr(codes), len(codes), C.byref(p), int(batch), cb, None)
        if failure:
            raise failure[0]
        check(rc, self.ctx._h)


class Hits:
    """Result of one search (vsc_hits): records sorted by (guide, strand, contig, pos)."""

    def __init__(self, genome, handle, codes, owned=True):
        self.genome, self._h, self.codes = genome, handle, codes
        self.ctx = genome.ctx
        self._owned = owned  # a batch of search_streamed belongs to the library: never freed from here
        if owned:
            self.ctx._children.add(self)

    def __len__(self):
        return int(lib().vsc_hits_count(self._h))

    @property
    def device_ptr(self):
        return lib().vsc_hits_data_dev(self._h)

    def to_numpy(self):
        n = len(self)
        p = C.c_void_p()
        check(lib().vsc_hits_data(self._h, C.byref(p)), self.ctx._h)
        if n == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        buf = (C.c_char * (n * HIT_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=HIT_DTYPE).copy()

    def copy_to(self, dst_ptr, dst_is_device):
        """Copy the records to caller memory (e.g. the data_ptr() of a uint8 tensor handed to RCCL)."""
        check(lib().vsc_hits_copy(self._h, C.c_void_p(dst_ptr), int(bool(dst_is_device))), self.ctx._h)

    def pack_exchange(self, dst_ptr, dst_is_device):
        """vsc_hits_pack_exchange: the records as 8-byte exchange records into caller memory (len(self) * 8 bytes,
        e.g. the data_ptr() of a uint8 tensor handed to RCCL); returns the per-key record counts
        (uint32[2 * reads], key = read << 1 | strand)."""
        counts = np.zeros(2 * len(self.codes), dtype=np.uint32)
        check(lib().vsc_hits_pack_exchange(self.ctx._h, self.genome._h, self._h, len(self.codes), C.c_void_p(dst_ptr),
                                           int(bool(dst_is_device)), ptr(counts)), self.ctx._h)
        return counts

    def scores(self, first=0, count=None, mit=True, features=False):
        """(mit float64[count] | None, mit_flags uint8[count] | None, features uint8[count,442] | None)."""
        count = len(self) - first if count is None else count
        m = np.empty(count, dtype=np.float64) if mit else None
        fl = np.empty(count, dtype=np.uint8) if mit else None
        ft = np.empty((count, N_FEATURES), dtype=np.uint8) if features else None
        check(lib().vsc_score_hits(self.genome.ctx._h, self.genome._h, self._h, ptr(self.codes), len(self.codes),
                                   first, count, ptr(m), ptr(fl), ptr(ft)), self.genome.ctx._h)
        return m, fl, ft

    def packed_features(self, first=0, count=None, to_host=True, mit=False, dev_ptr=None):
        """vsc_score_hits_packed: 64-byte feature rows (uint32[count, 16]) and optionally MIT scores.
        dev_ptr: device memory for count * 64 bytes (e.g. a torch tensor's data_ptr()) that receives the rows;
        without it they pass through library scratch (to_host=False: computed and dropped - timing runs)."""
        count = len(self) - first if count is None else count
        rows = np.empty((count, 16), dtype=np.uint32) if to_host else None
        m = np.empty(count, dtype=np.float64) if mit else None
        check(lib().vsc_score_hits_packed(self.genome.ctx._h, self.genome._h, self._h, ptr(self.codes), len(self.codes),
                                          first, count, C.c_void_p(dev_ptr) if dev_ptr else None, ptr(rows), ptr(m)),
              self.genome.ctx._h)
        return rows, m

    def close(self):
        if self._h:
            if self._owned:
                lib().vsc_hits_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MergedHits(Hits):
    """Result of vsc_hits_merge: the records of all genome shards in global order (rank 0 only)."""

    def __init__(self, ctx, handle):
        self.ctx, self._h, self.codes, self.genome = ctx, handle, None, None
        self._owned = True
        ctx._children.add(self)

    def to_numpy(self):
        n = len(self)
        out = np.zeros(n, dtype=HIT_DTYPE)
        if n:
            check(lib().vsc_hits_copy(self._h, ptr(out), 0), self.ctx._h)
        return out

    def scores(self, *a, **k):
        raise NotImplementedError("score hits on the rank that owns the shard, before the gather")


def merge_shard_records(ctx, records_ptr, on_device, shard_counts, n_guides):
    """records_ptr: pointer to the vsc_hit records of all shards concatenated in shard order
    (shard_counts[s] each; device memory if on_device, e.g. the buffer RCCL gathered into)."""
    counts = np.ascontiguousarray(shard_counts, dtype=np.uint64)
    h = C.c_void_p()
    check(lib().vsc_hits_merge(ctx._h, C.c_void_p(records_ptr), int(bool(on_device)), ptr(counts), len(counts), n_guides,
                               C.byref(h)), ctx._h)
    return MergedHits(ctx, h)


def merge_packed_records(ctx, genome, records_ptr, on_device, key_counts, first_key=0, votes_ptr=None, votes_out_ptr=None,
                         votes_out_on_device=False):
    """vsc_hits_merge_packed: records_ptr = the 8-byte exchange records of all shards concatenated in shard order,
    key_counts = uint32[n_shards, n_keys] (records of shard s with key first_key + k); `genome` = any shard of the
    genome on `ctx` (contig table).  Returns the merged vsc_hit records (guide = key >> 1).
    votes_ptr / votes_out_ptr (vsc_hits_merge_packed_votes): one uint16 per exchange record, in the memory space of the
    records, that travelled with them (the votes of the shard's classifier) -> the same values in merged order."""
    kc = np.ascontiguousarray(key_counts, dtype=np.uint32)
    assert kc.ndim == 2
    h = C.c_void_p()
    if votes_ptr is None:
        check(lib().vsc_hits_merge_packed(ctx._h, genome._h, C.c_void_p(records_ptr), int(bool(on_device)), ptr(kc), kc.shape[0],
                                          int(first_key), kc.shape[1], C.byref(h)), ctx._h)
    else:
        check(lib().vsc_hits_merge_packed_votes(ctx._h, genome._h, C.c_void_p(records_ptr), C.c_void_p(votes_ptr), int(bool(on_device)),
                                                ptr(kc), kc.shape[0], int(first_key), kc.shape[1], C.byref(h), C.c_void_p(votes_out_ptr),
                                                int(bool(votes_out_on_device))), ctx._h)
    return MergedHits(ctx, h)


class MultiContext:
    """vsc_multi: the genome-sharded search of ONE process over several devices behind the C ABI (one context per
    entry of `devices`; ids may repeat - several contexts on one GPU).  search() returns the merged records on
    the first context."""

    def __init__(self, devices, rccl=None, rccl_library=None):
        """rccl / rccl_library: the hooks of vsc_multi_create_debug (None: vsc_multi_create) - rccl=False forces
        device copies, rccl=True insists on RCCL (also with one device), rccl="try" attempts it and falls back, rccl_library names the one library to load."""
        ids = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        if rccl is None and rccl_library is None:
            check(lib().vsc_multi_create(ids, len(devices), C.byref(self._h)))
        else:
            p = _lib.MultiDebugParams(-1 if rccl is None else (2 if rccl == "try" else int(bool(rccl))),
                                      rccl_library.encode() if rccl_library else None)
            check(lib().vsc_multi_create_debug(ids, len(devices), C.byref(p), C.byref(self._h)))
        self.devices = list(devices)
        self._genomes = weakref.WeakSet()
        self._results = weakref.WeakSet()

    def _check(self, code):
        if code != 0:
            raise _lib.VarscotError(code, lib().vsc_multi_last_error(self._h).decode(errors="replace"))

    @property
    def uses_rccl(self):
        return bool(lib().vsc_multi_uses_rccl(self._h))

    def last_error(self):
        return lib().vsc_multi_last_error(self._h).decode(errors="replace")

    def set_debug(self, **hooks):
        """Context.set_debug on every context of the set."""
        d = _lib.DebugParams.defaults()
        for k, v in hooks.items():
            setattr(d, k, int(v))
        for i in range(len(self.devices)):
            check(lib().vsc_ctx_set_debug_params(lib().vsc_multi_ctx(self._h, i), C.byref(d) if hooks else None))

    def release_scratch(self):
        """vsc_multi_release_scratch: the pooled scratch of every context and the exchange buffers back to the devices."""
        self._check(lib().vsc_multi_release_scratch(self._h))

    def timing(self):
        t = _lib.MultiTiming()
        self._check(lib().vsc_multi_get_timing(self._h, C.byref(t)))
        return t.as_dict()

    def load_genome(self, packed):
        g = MultiGenome(self, packed)
        self._genomes.add(g)
        return g

    def close(self):
        if self._h:
            for r in list(self._results):
                r.close()
            for g in list(self._genomes):
                g.close()
            lib().vsc_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedContext:
    """The first context of a MultiContext as far as a result object needs it (owned by the vsc_multi)."""

    def __init__(self, handle):
        self._h = handle
        self._children = weakref.WeakSet()


class MultiGenome:
    def __init__(self, multi, packed):
        self.multi, self.packed = multi, packed
        self._h = C.c_void_p()
        hi, lo, nm = (np.ascontiguousarray(a, dtype=np.uint32) for a in (packed.hi, packed.lo, packed.nmask))
        contigs = np.ascontiguousarray(packed.contigs, dtype=CONTIG_DTYPE)
        multi._check(lib().vsc_multi_genome_load(multi._h, ptr(hi), ptr(lo), ptr(nm), len(hi), ptr(contigs), len(contigs),
                                                 C.byref(self._h)))

    def build_index(self, extra_pam=None):
        p = Genome._params(0, extra_pam, ALGO_SEED)
        self.multi._check(lib().vsc_multi_genome_build_index(self.multi._h, self._h, C.byref(p)))

    def search(self, guides, max_mismatches, extra_pam=None, algorithm="auto"):
        codes = guides if isinstance(guides, np.ndarray) else pack_guides(guides)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        p = Genome._params(max_mismatches, extra_pam, algorithm)
        h = C.c_void_p()
        self.multi._check(lib().vsc_multi_search(self.multi._h, self._h, ptr(codes), len(codes), C.byref(p), C.byref(h)))
        res = MergedHits(_BorrowedContext(C.c_void_p(lib().vsc_multi_result_ctx(self.multi._h))), h)
        self.multi._results.add(res)
        return res

    def search_streamed(self, guides, max_mismatches, on_batch, batch=0, extra_pam=None, algorithm="auto", score=None,
                        forest=None, guide_activity=None):
        """vsc_multi_search_stream: the reads go through all shards in batches of `batch`; what `score` names is computed
        per hit on the shard that found it, before the exchange ("rows": the packed feature rows, computed and dropped;
        "votes": `forest` walked per hit with the reads' `guide_activity`, 2 bytes per hit travel with the record); every
        merged batch is handed to on_batch(hits, first_guide, n_guides, votes_dev) - votes_dev: device address (first
        device) of one uint16 per record, or None - and freed afterwards.  The exchange and merge of a batch run while the
        shards search the next one."""
        codes = guides if isinstance(guides, np.ndarray) else pack_guides(guides)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        p = Genome._params(max_mismatches, extra_pam, algorithm)
        sc, keep = None, []
        if score:
            sc = _lib.MultiScore()
            sc.mode = {"rows": _lib.MULTI_SCORE_ROWS, "votes": _lib.MULTI_SCORE_VOTES}[score]
            if score == "votes":
                act = np.ascontiguousarray(guide_activity, dtype=np.float64)
                assert len(act) == len(codes)
                model = _lib.RfModel(forest.n_trees, forest.n_nodes, ptr(forest.status), ptr(forest.feature), ptr(forest.left),
                                     ptr(forest.right), ptr(forest.split), ptr(forest.node_class))
                keep += [act, model]
                sc.guide_activity = act.ctypes.data
                sc.model = C.pointer(model)
        failure = []
        rctx = _BorrowedContext(C.c_void_p(lib().vsc_multi_result_ctx(self.multi._h)))

        def trampoline(_user, handle, first, count, votes_dev):
            try:
                h = MergedHits(rctx, C.c_void_p(handle))
                h._owned = False  # the batch belongs to the library
                try:
                    on_batch(h, int(first), int(count), votes_dev)
                finally:
                    h.close()
                return 0
            except BaseException as e:  # no exception may cross the C boundary
                failure.append(e)
                return -5

        cb = _lib.MULTI_BATCH_FN(trampoline)
        rc = lib().vsc_multi_search_stream(self.multi._h, self._h, ptr(codes), len(codes), C.byref(p), int(batch),
                                           C.byref(sc) if sc is not None else None, cb, None)
        if failure:
            raise failure[0]
        self.multi._check(rc)

    def close(self):
        if self._h:
            lib().vsc_multi_genome_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LazyNames:
    """The ids of a variant-window genome (millions of them), decoded from the library's name pool on demand."""

    def __init__(self, pool, offsets):
        self._pool, self._off = pool, offsets

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self._pool[int(self._off[i]):int(self._off[i + 1]) - 1].tobytes().decode()

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def variant_windows(reference, vcf_path, sample=0, seq_len=23, threads=0):
    """vsc_windows_build: the alt-allele windows of one VCF sample column as a PackedGenome (the "SNP genome"
    of VARSCOT:296-307), built straight from the reference's packed planes - no FASTA in between."""
    L = lib()
    names = (C.c_char_p * len(reference.names))(*[n.encode() for n in reference.names])
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    code = L.vsc_windows_build(str(vcf_path).encode(), sample, seq_len, threads, ptr(reference.hi), ptr(reference.lo),
                               ptr(reference.nmask), ptr(reference.contigs), names, len(reference.contigs), C.byref(h), err, 512)
    if code != 0:
        raise _lib.VarscotError(code, err.value.decode(errors="replace"))
    owner = _WindowsOwner(h)  # the arrays below are views of the library's memory: it lives as long as they do
    n, nw = int(L.vsc_windows_count(h)), int(L.vsc_windows_words(h))

    def view(address, ctype, count, dtype):
        a = np.ctypeslib.as_array(C.cast(address, C.POINTER(ctype)), shape=(count,)).view(dtype)
        return _OwnedArray.wrap(a, owner)

    out = PackedGenome.__new__(PackedGenome)
    out.hi, out.lo, out.nmask = (view(L.vsc_windows_plane(h, k), C.c_uint32, nw, np.uint32) for k in range(3))
    if n:
        out.contigs = view(L.vsc_windows_contigs(h), C.c_uint8, n * CONTIG_DTYPE.itemsize, CONTIG_DTYPE)
        offsets = view(L.vsc_windows_name_offsets(h), C.c_uint64, n + 1, np.uint64)
        pool = view(L.vsc_windows_name(h, 0, None), C.c_uint8, int(offsets[n]), np.uint8)
        out.names = LazyNames(pool, offsets)
    else:
        out.contigs, out.names = np.zeros(0, dtype=CONTIG_DTYPE), []
    return out


class _WindowsOwner:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                lib().vsc_windows_free(self._h)
                self._h = None
        except Exception:
            pass


class _OwnedArray(np.ndarray):
    """ndarray view that keeps the owner of its memory alive."""

    @classmethod
    def wrap(cls, a, owner):
        v = a.view(cls)
        v._owner = owner
        return v

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def unpack_features(rows):
    """Packed 64-byte feature rows -> dense uint8[n, 442] (vsc_unpack_features)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint32).reshape(-1, 16)
    out = np.empty((len(rows), N_FEATURES), dtype=np.uint8)
    lib().vsc_unpack_features(ptr(rows), len(rows), ptr(out))
    return out


def sam_order(hits):
    """(order, secondary) in which read_mapping/bidir_mapping.cpp:167-187 writes a sorted result."""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    order = np.empty(len(hits), dtype=np.uint64)
    sec = np.empty(len(hits), dtype=np.uint8)
    lib().vsc_sam_order(ptr(hits), len(hits), ptr(order), ptr(sec))
    return order, sec


def device_count():
    return int(lib().vsc_device_count())
