"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm), the
genome sharded by tile-aligned plane ranges, every rank searching all reads on its shard, and ONE
exchange of hit records per search over xGMI:

  exchange="root"   the gather of all records to rank 0 (one consumer; rank 0's links carry everything:
                    (N-1)/N of the result over N-1 links)
  exchange="reads"  every rank gathers the hits of ITS read range from all genome shards (N gathers at
                    once = an all-to-all; each link carries 1/N^2 of the result) and merges them; the
                    result stays distributed, sorted, in read-range order

torch is used for the process group and the device buffers that RCCL moves - nothing else.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import TILE_WORDS, HIT_DTYPE, merge_shard_records

RECORD_BYTES = HIT_DTYPE.itemsize


def shard_words(n_words_total, rank, world):
    """Tile-aligned word range [begin, end) of the planes owned by `rank` (same rule as
    PackedGenome.shard_words; windows are owned by the shard that holds their first base)."""
    tiles = (n_words_total + TILE_WORDS - 1) // TILE_WORDS
    b = (tiles * rank // world) * TILE_WORDS
    e = (tiles * (rank + 1) // world) * TILE_WORDS
    return min(b, n_words_total), min(e, n_words_total)


def gather_records(local, group=None, dst=0):
    """local: 1-D uint8 tensor holding this rank's records (device tensor for nccl, CPU for gloo).

    Returns (on dst) a uint8 tensor with the records of rank 0, 1, ... concatenated in rank order and
    the per-rank record counts; (None, counts) elsewhere.  One all_gather of the counts (8 bytes per
    rank) + one grouped send/recv of the variable-length payloads straight into their final place.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.numel() // RECORD_BYTES], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        out = torch.empty(sum(counts) * RECORD_BYTES, dtype=torch.uint8, device=local.device)
        offs = np.concatenate([[0], np.cumsum(counts)]) * RECORD_BYTES
        out[offs[dst]:offs[dst + 1]] = local
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group)
               for r in range(world) if r != dst and counts[r] > 0]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, local, dst, group)] if counts[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def read_range(n_reads, rank, world):
    """Reads [begin, end) whose hits `rank` collects under exchange="reads"."""
    return n_reads * rank // world, n_reads * (rank + 1) // world


def start_exchange_by_reads(local, n_reads, group=None):
    """Issues the exchange of exchange_by_reads and returns without waiting for the payloads:
    (receive buffer, per-source counts, outstanding requests)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rec = local.view(torch.int32).view(-1, 4)
    bounds = torch.tensor([read_range(n_reads, d, world)[0] for d in range(world)] + [n_reads], dtype=torch.int32,
                          device=local.device)
    cut = torch.searchsorted(rec[:, 0].contiguous(), bounds)  # first record of every destination's read range
    send = (cut[1:] - cut[:-1]).to(torch.int64)
    rows = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(rows, send, group=group)
    matrix = torch.stack(rows).cpu().numpy()  # matrix[s][d] = records s sends to d
    cut = cut.cpu().numpy().astype(np.int64) * RECORD_BYTES
    counts = [int(matrix[s][rank]) for s in range(world)]
    out = torch.empty(sum(counts) * RECORD_BYTES, dtype=torch.uint8, device=local.device)
    offs = np.concatenate([[0], np.cumsum(counts)]) * RECORD_BYTES
    out[offs[rank]:offs[rank + 1]] = local[cut[rank]:cut[rank + 1]]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        if matrix[rank][peer] > 0:
            ops.append(dist.P2POp(dist.isend, local[cut[peer]:cut[peer + 1]], peer, group))
        if counts[peer] > 0:
            ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], peer, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return out, counts, reqs


def exchange_by_reads(local, n_reads, group=None):
    """local: 1-D uint8 tensor with this rank's records, sorted by (read, strand, contig, pos).

    Every rank receives, from every rank, the records of its own read range (read_range); returns the
    received records concatenated in source-rank order (= genome-shard order, what vsc_hits_merge
    expects) and the per-source counts.  One all_gather of the world x world count matrix + one
    grouped send/recv of the payloads straight into their final place."""
    out, counts, reqs = start_exchange_by_reads(local, n_reads, group)
    for req in reqs:
        req.wait()
    return out, counts


class _DeviceAlias:
    """Zero-copy view of library-owned device memory for torch (the records of a vsc_hits)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _records_tensor(hits, device):
    """The records of `hits` as a uint8 tensor on `device` (a CPU copy for gloo, an alias for nccl)."""
    n = len(hits)
    if device is not None and torch.device(device).type == "cuda" and n:
        try:
            return torch.as_tensor(_DeviceAlias(hits.device_ptr, n * RECORD_BYTES), device=device)
        except (TypeError, RuntimeError, ValueError):
            pass  # a torch build without __cuda_array_interface__ import: stage a copy instead
    local = torch.empty(n * RECORD_BYTES, dtype=torch.uint8, device=device)
    if n:
        hits.copy_to(local.data_ptr(), local.is_cuda)
    return local


def sharded_search(ctx, genome_shard, codes, max_mismatches, extra_pam=None, group=None, device=None,
                   algorithm="auto", exchange="root"):
    """Search all reads on this rank's shard, exchange the hit records once, merge.

    exchange="root":  returns (merged hits of all reads on rank 0 | None elsewhere, local Hits)
    exchange="reads": returns (merged hits of this rank's read range, local Hits)"""
    hits = genome_shard.search(codes, max_mismatches, extra_pam, algorithm=algorithm)
    local = _records_tensor(hits, device)
    if exchange == "reads":
        gathered, counts = exchange_by_reads(local, len(codes), group)
    elif exchange == "root":
        gathered, counts = gather_records(local, group)
    else:
        raise ValueError("exchange must be 'root' or 'reads'")
    merged = None
    if gathered is not None:
        if gathered.is_cuda:
            torch.cuda.current_stream(gathered.device).synchronize()
        merged = merge_shard_records(ctx, gathered.data_ptr(), gathered.is_cuda, counts, len(codes))
    return merged, hits


def sharded_search_pipelined(ctx, genome_shard, codes, max_mismatches, extra_pam=None, group=None, device=None,
                             algorithm="auto", sub_batches=4, timings=None):
    """exchange="reads" with the read set cut into `sub_batches` consecutive pieces: the exchange of piece i
    travels over xGMI while piece i + 1 is searched ("chunked by guide batch so that
    gather overlaps the next batch's scan").

    Returns [(first read of the piece, merged hits of this rank's share of the piece)], in read order; the
    records' `guide` fields count from the first read of their piece.  `timings`, if a list, receives the
    library's timing of every piece's search."""
    n = len(codes)
    cuts = [n * i // sub_batches for i in range(sub_batches + 1)]
    pieces = [(cuts[i], codes[cuts[i]:cuts[i + 1]]) for i in range(sub_batches) if cuts[i + 1] > cuts[i]]
    out = []
    pending = None

    def finish(p):
        first, n_reads, hits, local, recv, counts, reqs = p
        for req in reqs:
            req.wait()
        if recv.is_cuda:
            torch.cuda.current_stream(recv.device).synchronize()
        out.append((first, merge_shard_records(ctx, recv.data_ptr(), recv.is_cuda, counts, n_reads)))
        hits.close()  # the send buffers alias its records: only now

    for first, part in pieces:
        hits = genome_shard.search(part, max_mismatches, extra_pam, algorithm=algorithm)  # overlaps the pending exchange
        if timings is not None:
            timings.append(dict(ctx.timing()))
        if pending is not None:
            finish(pending)
        local = _records_tensor(hits, device)
        recv, counts, reqs = start_exchange_by_reads(local, len(part), group)
        pending = (first, len(part), hits, local, recv, counts, reqs)
    if pending is not None:
        finish(pending)
    return out
