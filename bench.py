#!/usr/bin/env python3
"""bench.py - headline benchmark of the off-target search hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without a launcher's environment it starts
                                                              torch.distributed.run itself, as a child process, before
                                                              anything touches a GPU, and relays the child's JSON line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: all reads of the workload
searched (both strands) against the resident synthetic 3 Gbp genome - search kernel, bin sort of the
hits, record assembly - and, for N > 1, ONE exchange of 8-byte hit records over RCCL plus the merge:
both shapes are timed and reported (`exchanges`): "root" = the single gather to rank 0 (north star; the headline `value`),
"reads" = every rank gathers and merges its read range (result stays distributed).  Inputs (packed genome planes) are
resident in HBM before the timed region starts; results stay in HBM.  The genome is sharded by position range across ranks
(total work fixed => strong scaling).  Rank 0 prints ONE JSON line.

Two multi-GPU drivers share the kernels and the exchange record:
  --multi torch   one process per GPU over torch.distributed / RCCL (varscot_amd/dist.py) - what the launcher contract starts
  --multi abi     ONE process over the N devices behind the C ABI (vsc_multi_*: what `bidir_mapping -D 0,..,N-1` executes)
  --multi both    (default for N > 1) the torch path is the headline; rank 0 first runs the abi path as a child process - before
                  it touches a GPU itself - and embeds that line as `multi_abi`
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (reads, genome bases, max mismatches, description)  - BASELINE.json configs
    "c1": (10, 1_000_000, 4, "10 guides, 1 Mbp synthetic genome, <=4 mismatches"),
    "c2": (1_000, 3_000_000_000, 6, "1 000 guides, hg38-sized 3 Gbp synthetic ref, <=6 mismatches"),
    "c3": (10_000, 3_000_000_000, 8, "10 000 guides, 3 Gbp synthetic ref, <=8 mismatches, genome-sharded"),
    "c4": (1_000, 3_000_000_000, 6, "1 000 guides, 3 Gbp synthetic ref + synthetic VCF (~5 M SNPs): reference and "
                                    "alt-allele windows (SNP genome) searched, <=6 mismatches, 1 GPU"),
    "c5": (100_000, 3_000_000_000, 8, "100 000 guides streamed in batches, 3 Gbp ref, <=8 mismatches + packed per-hit "
                                      "feature rows and MIT scores"),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz (32-bit integer lane-ops / s)
LANE_OPS_PER_COMPARE = {"scan": 3.5,     # v_xor + v_bitop3 + v_bcnt + 1/2 v_min3 per (site, read) pair
                        "sliced": 1.8125}  # 58 instructions per read and 32 sites: 14 x 2 mismatch vectors (the PAM positions are the
                                           # chunk's class) + 22 adder tree + 5 test + 3 duplicate test (0 / 3 / 6 by segment)
# measured issue cost of a vector instruction (tools/micro/valu_rate.hip -> profiles/r03_valu_rate_microbench.txt, cycles per
# SIMD at 2.4 GHz with six waves resident): 2.6 with vector operands only, 4.2 with one scalar operand
VALU_ISSUE_CYCLES = {"vector_operands": 2.6, "scalar_operand": 4.2}
KERNEL_SOURCES = ("vsc_seed.hip", "vsc_kernels.hip", "vsc_sort.hip", "vsc_device.h", "vsc_internal.h")


# pair-node forest walk: 15.2 vector instructions per step (2 levels), ~40 SIMD-cycles at the costs of tools/micro/valu_kinds.hip
FOREST_CYCLES_PER_INSTRUCTION = 2.6


def kernel_sources_sha():
    """What the committed counters under profiles/ are keyed by: a hash of the kernels' CODE - comments, blank lines and
    trailing blanks aside, so that rewording a comment does not orphan a profile.  A lookup made with other kernels than the
    ones that were profiled prints null instead of a stale number."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "varscot_amd", "csrc", name), "r") as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # (no string literal of these files holds a comment marker)
        lines = [re.sub(r"//.*", "", ln).rstrip() for ln in text.split("\n")]
        h.update(name.encode() + b"\0" + "\n".join(ln for ln in lines if ln.strip()).encode())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--guides", type=int, default=None, help="override the read count")
    ap.add_argument("--bases", type=int, default=None, help="override the genome size")
    ap.add_argument("--mismatches", type=int, default=None)
    ap.add_argument("--algorithm", default="auto", choices=["auto", "scan", "seed"],
                    help="scan = stream the packed planes; seed = resident pigeonhole site tables; auto = seed")
    ap.add_argument("--snps", type=int, default=5_000_000, help="records of the synthetic VCF (workload c4)")
    ap.add_argument("--batch", type=int, default=10_000, help="reads per search call for the streamed workload c5")
    ap.add_argument("--sub-batches", type=int, default=None,
                    help="multi-rank runs: cut the reads into this many pieces and overlap the exchange of one piece with the "
                         "search of the next (default: --exchange root 4; --exchange reads 4 at 2 ranks, 2 at 3-4, 1 otherwise)")
    ap.add_argument("--rehearse", action="store_true",
                    help="rehearsal of the multi-rank control flow on a box with ONE GPU: all ranks use cuda:0, the "
                         "process group is gloo and the records travel through host memory (use a small workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange/merge path even with one rank")
    ap.add_argument("--exchange", default="root", choices=["reads", "root"],
                    help="multi-rank exchange of the hit records that `value` reports: 'root' = everything to rank 0 (the north "
                         "star's single gather), 'reads' = every rank gathers and merges its read range (all-to-all, result stays "
                         "distributed); both are timed and reported")
    ap.add_argument("--multi", default=None, choices=["torch", "abi", "both"],
                    help="multi-GPU driver: torch = one process per GPU (torch.distributed / RCCL), abi = ONE process over the "
                         "devices behind the C ABI (vsc_multi_*), both = torch as the headline + the abi line embedded (default "
                         "for N > 1)")
    ap.add_argument("--abi-devices", default=None,
                    help="--multi abi: the device list, e.g. 0,0,0,0 (ids may repeat: a rehearsal of N shards on one GPU); "
                         "default 0..gpus-1")
    ap.add_argument("--cpu-sample-bases", type=int, default=192_000_000)
    ap.add_argument("--master-port", type=int, default=None, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--classify", action="store_true",
                    help="workload c5: score -> classify fused (vsc_score_classify_hits: the reference's forest walked per hit, "
                         "2 bytes of votes per hit out) instead of writing the 64-byte packed feature rows")
    ap.add_argument("--rows", default="fused", choices=["fused", "two-pass"],
                    help="workload c5 on one GPU: fused = vsc_search_stream_rows (the search keeps the sites' bases beside the records, "
                         "the record assembly writes the feature rows); two-pass = vsc_search_stream + vsc_score_hits_packed from the "
                         "callback (re-reads the records, gathers the windows from the planes)")
    ap.add_argument("--hook", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment hook of include/varscot_hip_debug.h (e.g. seed_groups_per_cu=2); the defaults are what "
                         "every reported number uses")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch check without a GPU: the ranks form the process group (gloo with --rehearse), all-reduce "
                         "once, rank 0 prints a JSON line saying what the communicator reported")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one rank per GPU with
    torch.distributed.run as a CHILD process (this process has not touched a GPU and never will), relay the child's
    JSON line and exit with its code."""
    import socket
    import subprocess
    # A profiler's preloaded library (rocprofv3, always with --pmc) has initialised the GPU before this program started:
    # starting the launcher from here would be an exec from a GPU-initialised process, which takes a machine of this pool
    # down.  Multi-rank profiling starts the launcher first and the profiler per rank, not the other way round.
    preload = os.environ.get("LD_PRELOAD", "")
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_TOOL")) for k in os.environ) or "rocprofiler" in preload or "rocprof" in preload:
        print("bench.py: --gpus %d under a profiler: refusing to start the ranks from a process the profiler has already "
              "attached to (run `python -m torch.distributed.run ... bench.py` and profile inside the ranks)" % args.gpus, file=sys.stderr)
        raise SystemExit(2)
    port = args.master_port
    if port is None:
        with socket.socket() as sck:
            sck.bind(("127.0.0.1", 0))
            port = sck.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in child.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    raise SystemExit(child.returncode if child.returncode or lines else 1)


def cpu_baseline(total_bases, max_mm, sample_bases, target_s=7.0):
    """The reference's CPU path cannot be built here (every translation unit needs SeqAn), so the baseline is
    the oracle's two ports (kind "port"), each timed on a bounded sample of the same synthetic genome and read
    set with all host threads, for at least ~5 s of wall time (more reads, not more bases), scaled linearly in
    the genome length to the full reference:
      * pigeonhole-flow port (oracle/vsc_pigeon.c): the reference's algorithm shape - halves with floor(m/2)
        substitutions through a k-mer table of the slice, every occurrence verified by the delegate, OpenMP over
        reads (read_mapping/bidir_mapping.cpp:129-162,285-295); the table build is index construction, untimed
      * bit-parallel scan port (oracle/vsc_fastport.c): every PAM-valid window against every read.
    `value` is the faster of the two."""
    from oracle import pyoracle
    from varscot_amd import synth
    import ctypes as C
    from varscot_amd._lib import lib, ptr
    pyoracle.build()
    threads = pyoracle.max_threads()
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"

    def slice_text(n_bases):
        w0 = 20_000 // 32  # past the leading N block of chr1
        hi, lo, nm, _, _, _ = synth.synthetic_planes(total_bases, w0, w0 + n_bases // 32)
        buf = C.create_string_buffer(len(hi) * 32)
        lib().vsc_unpack_bases(ptr(hi), ptr(lo), ptr(nm), 0, len(hi) * 32, buf)
        return buf.raw

    def timed(run, n_probe, n_max):
        """run(n_reads) -> (hits, work); a probe sizes the real run to about target_s seconds"""
        t0 = time.perf_counter()
        run(n_probe)
        probe = max(time.perf_counter() - t0, 1e-3)
        n = int(min(n_max, max(n_probe, n_probe * target_s / probe)))
        t0 = time.perf_counter()
        hits, work = run(n)
        return n, time.perf_counter() - t0, hits, work

    entries = []
    # ---- pigeonhole-flow port ----
    text = slice_text(min(sample_bases, 48_000_000))
    ix = pyoracle.PigeonIndex([text])
    _, reads = synth.synthetic_guides(16384)
    n, dt, hits, cand = timed(lambda k: ix.count(reads[:k], max_mm, threads=threads), 2 * threads, len(reads))
    ix.close()
    scale = len(text) / float(total_bases)
    entries.append({"name": "pigeonhole-flow port (oracle/vsc_pigeon.c): k-mer table + verify, the reference's algorithm shape",
                    "value": n / dt * scale, "unit": "guides/s", "reads": n, "slice_bases": len(text), "wall_s": dt,
                    "sites_per_s": hits / dt, "delegate_calls": cand})
    # ---- bit-parallel scan port ----
    text = slice_text(sample_bases)
    n, dt, hits, sites = timed(lambda k: pyoracle.count_fast([text], reads[:k], max_mm, threads=threads), 64, 4096)
    scale = len(text) / float(total_bases)
    entries.append({"name": "bit-parallel scan port (oracle/vsc_fastport.c): every PAM-valid window x every read",
                    "value": n / dt * scale, "unit": "guides/s", "reads": n, "slice_bases": len(text), "wall_s": dt,
                    "sites_per_s": hits / dt, "windows_compared": sites})
    best = max(entries, key=lambda e: e["value"])
    return {
        "value": best["value"], "unit": "guides/s", "cores": threads, "kind": "port", "cpu_model": model,
        "sample": "two CPU ports of the oracle on slices of the same synthetic genome and read set, <=%d mismatches, all %d host "
                  "threads, %.1f s and %.1f s wall, scaled linearly in genome length to %.1f Gbp (the reference's FM-index "
                  "search needs SeqAn and cannot be built here); value = the faster one: %s"
                  % (max_mm, threads, entries[0]["wall_s"], entries[1]["wall_s"], total_bases / 1e9, best["name"].split(" (")[0]),
        "entries": entries,
    }


def sort_roofline(timing_sums, steps):
    """The bin sort (vsc_sort.hip): bytes its kernels moved (vsc_timing.sort_bytes: 8 per packed record read or
    written, 16 per result record) over the time from the end of the search kernel to the last result record."""
    ms = (timing_sums["sort_ms"] + timing_sums["finalize_ms"]) / steps
    return {"bound": "hbm", "kernel": "bin_hist + bin_partition + bin_finalize (hand-written, vsc_sort.hip)",
            "achieved": timing_sums["sort_bytes"] / steps / (max(ms, 1e-9) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "launch_ms": ms, "partition_levels": timing_sums["sort_levels"], "first_level_bin_bits": timing_sums["sort_bin_bits"],
            "bytes": timing_sums["sort_bytes"] / steps}


def run_abi_child(args):
    """--multi both: the one-process-over-N-devices line (`bench.py --multi abi`) measured by a CHILD process that rank 0
    starts before it has touched a GPU itself, so that the two drivers never share a process or a moment on the devices.
    Returns the child's parsed line, or {"error": ...} - the headline does not depend on it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--multi", "abi", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--workload", args.workload, "--algorithm", args.algorithm, "--batch", str(args.batch),
           "--no-cpu-baseline"]
    for flag, v in (("--guides", args.guides), ("--bases", args.bases), ("--mismatches", args.mismatches)):
        if v is not None:
            cmd += [flag, str(v)]
    if args.classify:
        cmd.append("--classify")
    if args.abi_devices:
        cmd += ["--abi-devices", args.abi_devices]
    elif args.rehearse:  # a one-GPU box: N contexts on device 0
        cmd += ["--abi-devices", ",".join(["0"] * args.gpus)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    except Exception as e:  # noqa: BLE001
        return {"error": "child not run: %r" % (e,)}
    lines = [l for l in child.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if child.returncode != 0 or not lines:
        return {"error": "exit code %d: %s" % (child.returncode, child.stderr.decode(errors="replace")[-600:])}
    try:
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": "unparsable line: %r" % (e,)}


def main_abi(args, json_fd):
    """--multi abi: ONE process, vsc_multi over the devices (varscot_amd.MultiContext), K timed steps of vsc_multi_search
    (c2 / c3) or vsc_multi_search_stream with the per-hit scoring on the owning shard (c5) - what `bidir_mapping -D 0,1,...`
    executes.  Same line as the other driver: value = reads x steps / wall, hits from the library's own counters."""
    import varscot_amd as va
    from varscot_amd import synth
    n_guides, total_bases, max_mm, desc = WORKLOADS[args.workload]
    n_guides = args.guides or n_guides
    total_bases = args.bases or total_bases
    max_mm = args.mismatches if args.mismatches is not None else max_mm
    if args.workload == "c4":
        raise SystemExit("workload c4 is a single-GPU configuration")
    devices = [int(x) for x in args.abi_devices.split(",")] if args.abi_devices else list(range(args.gpus))
    if va.device_count() < 1 + max(devices):
        raise SystemExit("bench.py --multi abi: device %d asked for, %d visible" % (max(devices), va.device_count()))
    algorithm = "seed" if args.algorithm == "auto" else args.algorithm
    t_gen = time.perf_counter()
    packed = synth.synthetic_genome(total_bases)
    t_gen = time.perf_counter() - t_gen
    ids, seqs = synth.synthetic_guides(n_guides)
    codes = va.pack_guides(seqs)
    m = va.MultiContext(devices)
    g = m.load_genome(packed)
    del packed
    if algorithm == "seed":
        g.build_index()
    streamed = args.workload == "c5"
    forest, act = None, None
    if args.classify:
        from varscot_amd.classifier import Forest
        forest = Forest()
        act = np.random.default_rng(0x5EED0004).uniform(0.2, 1.8, size=n_guides)
    phases = dict.fromkeys(("search_wall_ms", "search_ms_max", "score_ms_max", "exchange_ms", "merge_ms", "callback_ms", "total_ms"), 0.0)
    state = {"hits": 0, "bytes": 0, "batches": 0}

    def step(timed):
        if streamed:
            seen = []
            g.search_streamed(codes, max_mm, lambda h, first, count, votes: seen.append(len(h)), batch=args.batch, algorithm=algorithm,
                              score=("votes" if forest is not None else "rows"), forest=forest, guide_activity=act)
            n = sum(seen)
        else:
            h = g.search(codes, max_mm, algorithm=algorithm)
            n = len(h)
            h.close()
        if timed:
            t = m.timing()
            for k in phases:
                phases[k] += t[k]
            state["hits"], state["bytes"], state["batches"] = n, t["exchanged_bytes"], t["batches"]

    for _ in range(args.warmup):
        step(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    dt = time.perf_counter() - t0   # (every call returns with all devices idle: the library synchronises its streams)
    out = {
        "metric": "guides/sec (candidate sites/sec alongside) at <=%d mismatches on a %.1f Gbp reference" % (max_mm, total_bases / 1e9),
        "value": n_guides * args.steps / dt, "unit": "guides/s", "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.workload, desc), "guides": n_guides, "genome_bases": total_bases, "max_mismatches": max_mm,
                   "parallelism": "genome-shard x%d" % len(devices), "algorithm": algorithm,
                   "multi_gpu_path": "ONE process over the devices behind the C ABI: vsc_multi_search%s (csrc/vsc_multi.cpp)"
                                     % ("_stream" if streamed else ""),
                   "devices": devices, "exchange": "root", "rccl_ranks": (len(devices) if m.uses_rccl else 0),
                   "exchange_transport": ("RCCL ncclSend / ncclRecv over xGMI" if m.uses_rccl else
                                          "device copies (%s)" % (m.last_error() or "repeated device ids / one device")),
                   "hits_per_step": int(state["hits"]), "candidate_sites_per_s": state["hits"] * args.steps / dt,
                   "batch": args.batch if streamed else n_guides, "batches": state["batches"],
                   "per_hit_scoring": (None if not streamed else "score -> classify fused on the owning shard, 2 B of votes per hit travel"
                                       if forest is not None else "64-byte packed feature rows on the owning shard (computed, dropped)")},
        "vsc_multi_timing": dict({k: v / args.steps for k, v in phases.items()}, exchanged_bytes=state["bytes"],
                                 note="host wall times per step (vsc_multi_timing): search = until the slowest shard was ready, "
                                      "exchange = what the transfers took beyond that, merge on the first device"),
        "setup": {"genome_generate_s": t_gen},
        "n_gt_1_rccl_executed": bool(m.uses_rccl and len(set(devices)) > 1),
    }
    g.close()
    m.close()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def main():
    args = parse()
    under_launcher = "WORLD_SIZE" in os.environ
    multi = args.multi or ("both" if args.gpus > 1 else "torch")
    if multi == "abi":
        # one process does everything: under a launcher that is rank 0, the others have nothing to do
        if under_launcher and int(os.environ.get("RANK", "0")) != 0:
            return
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        return main_abi(args, json_fd)
    if args.gpus > 1 and not under_launcher:
        launch_ranks(args)
    abi_line = None
    if multi == "both" and args.gpus > 1 and int(os.environ.get("RANK", "0")) == 0 and not args.dry_run:
        abi_line = run_abi_child(args)  # before this process touches a GPU
    # stdout carries exactly one line, the JSON result: everything else that might write to fd 1
    # (RCCL prints a banner there) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo" if args.rehearse or not torch.cuda.is_available() else "nccl", rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.float64, device="cuda:%d" % local_rank if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        if rank == 0:
            os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": args.gpus, "rccl_ranks": dist.get_world_size(),
                                           "backend": dist.get_backend(), "ranks_seen": int(one.item())}) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return
    import varscot_amd as va
    from varscot_amd import synth
    from varscot_amd import dist as vdist

    n_guides, total_bases, max_mm, desc = WORKLOADS[args.workload]
    n_guides = args.guides or n_guides
    total_bases = args.bases or total_bases
    max_mm = args.mismatches if args.mismatches is not None else max_mm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
    xdev = None if args.rehearse else device  # where the exchanged records live (None: host memory)

    # ---- inputs: this rank's shard of the synthetic genome, resident in HBM ------------------------
    table, names = synth.contig_table(total_bases)
    span = int(table[-1]["offset"]) + int(table[-1]["length"]) + 1
    n_words_total = (span + 31) // 32
    wb, we = vdist.shard_words(n_words_total, rank, world)
    t_gen = time.perf_counter()
    hi, lo, nm, _, _, _ = synth.synthetic_planes(total_bases, wb, min(we + 1, n_words_total))
    t_gen = time.perf_counter() - t_gen
    ctx = va.Context(local_rank)
    hooks = {h.split("=", 1)[0]: int(h.split("=", 1)[1]) for h in args.hook}
    if hooks:
        ctx.set_debug(**hooks)
    genome = va.Genome.from_shard(ctx, hi, lo, nm, wb, we - wb, table)
    del hi, lo, nm
    ids, seqs = synth.synthetic_guides(n_guides)
    codes = va.pack_guides(seqs)
    snp_genome, snp_info = None, None
    if args.workload == "c4":
        # variant-aware run: the alt-allele windows of the sample come straight from the reference's packed planes
        # (vsc_windows_build: VCF parsing, overlap sweep and window assembly on the host threads, no FASTA in
        # between) and are searched as a second resident genome.  Input preparation, outside the timed region;
        # the synthetic VCF itself is workload synthesis and not part of "prepare".
        if world != 1:
            raise SystemExit("workload c4 is a single-GPU configuration")
        import tempfile
        full = synth.synthetic_genome(total_bases)
        tmp = tempfile.mkdtemp(prefix="vsc_c4_")
        n_snps = synth.synthetic_vcf(full, args.snps, os.path.join(tmp, "in.vcf"))
        t_prep = time.perf_counter()
        snp_packed = va.variant_windows(full, os.path.join(tmp, "in.vcf"), sample=0)
        t_windows = time.perf_counter() - t_prep
        snp_genome = ctx.load_genome(snp_packed)
        snp_info = {"snps": n_snps, "windows": len(snp_packed.contigs), "window_bases": snp_packed.n_bases,
                    "windows_build_s": t_windows, "prepare_s": time.perf_counter() - t_prep,
                    "route": "vsc_windows_build: VCF -> packed planes (no FASTA round trip)"}
        del snp_packed, full
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    algorithm = "seed" if args.algorithm == "auto" else args.algorithm
    index_ms = None
    if algorithm == "seed":  # the resident site tables are part of the inputs, like the reference's FM index
        genome.build_index()
        index_ms = ctx.timing()["index_ms"]
        if snp_genome is not None:
            snp_genome.build_index()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    streamed = args.workload == "c5"
    rows_fused = streamed and args.rows == "fused" and not args.classify and world == 1 and not args.force_dist and algorithm == "seed"
    forest, guide_activity = None, None
    if args.classify:
        from varscot_amd.classifier import Forest
        forest = Forest()
        # synthetic on-target activities in the range of the reference's TUSCAN scores (workflow/guideseq-data/
        # guideseqOntargetActivity.txt: 0.2 .. 1.8), one per read, seeded
        guide_activity = np.random.default_rng(0x5EED0004).uniform(0.2, 1.8, size=n_guides)

    def score_batch(h, first, count):
        if forest is not None:
            # (a streamed batch numbers its reads globally; a per-batch search of the multi-rank path from 0)
            act = guide_activity if len(h.codes) == n_guides else guide_activity[first:first + count]
            forest.classify_hits(h, act, to_host=False)
        else:
            h.packed_features(to_host=False, mit=False)

    # more pieces = more of the exchange hidden, but every piece visits all site chunks again (c3 on one GPU:
    # 102 ms in one piece, 116 ms in four): 4 pieces at 2 ranks (13 GB over one link), 2 at 4, 1 at 8
    # (--exchange root: the gather to rank 0 is the long leg at every N - 13 GB into one GPU's links - so always 4 pieces:
    # piece i travels while piece i + 1 is searched and piece i - 1 is merged, varscot_amd.dist.sharded_search_stream)
    if args.sub_batches is not None:
        sub_batches = args.sub_batches
    else:
        sub_batches = {2: 4, 3: 2, 4: 2}.get(world, 1) if args.exchange == "reads" else 4
    pipelined = use_dist and sub_batches > 1 and args.workload in ("c1", "c2", "c3")
    # (the headline is --exchange root unless asked otherwise: the north star's single gather; "reads" is reported beside it)

    # per-step sums of the library's device timings (vsc_timing): every step function returns
    # (objects to close, records of this rank's final result, sums)
    T_SUM = ("scan_ms", "sort_ms", "finalize_ms", "prep_ms", "score_ms", "hits", "pairs", "genome_bytes", "sort_bytes", "list_entries")
    T_MAX = ("sort_levels", "sort_bin_bits", "passes", "sites", "read_passes", "seed_cut")

    def new_acc():
        return dict.fromkeys(T_SUM + T_MAX, 0)

    def add_timing(acc, t, score=False):
        for k in T_SUM:
            if k != "score_ms" or score:
                acc[k] += t[k]
        for k in T_MAX:
            acc[k] = max(acc[k], t[k])

    mode = {"exchange": args.exchange}   # which exchange the step functions use (both are timed when N > 1)
    X_KEYS = ("search_ms", "pack_ms", "exchange_ms", "merge_ms", "sent_bytes", "received_bytes")
    xacc = dict.fromkeys(X_KEYS, 0.0)    # host wall times / bytes of this rank's exchanges, summed over the timed steps

    def search_batch(batch_codes, acc):
        if not use_dist:
            h = genome.search(batch_codes, max_mm, algorithm=algorithm)
            add_timing(acc, ctx.timing())
            return h, None
        st = {}
        merged, local = vdist.sharded_search(ctx, genome, batch_codes, max_mm, device=xdev, algorithm=algorithm,
                                             exchange=mode["exchange"], stats=st)
        for k in X_KEYS:
            xacc[k] += st[k]
        add_timing(acc, ctx.timing())
        return local, merged

    def step_pipelined():
        timings, acc = [], new_acc()
        if mode["exchange"] == "root":
            total = [0]

            def on_piece(mh, first, count, votes):
                if mh is not None:
                    total[0] += len(mh)

            def note(h, first, count):          # (called after every piece's search on the rank that ran it)
                add_timing(acc, ctx.timing())
                return None

            vdist.sharded_search_stream(ctx, genome, codes, max_mm, on_piece, -(-n_guides // sub_batches), device=xdev,
                                        algorithm=algorithm, score=note)
            return [], total[0], acc
        parts = vdist.sharded_search_pipelined(ctx, genome, codes, max_mm, device=xdev, algorithm=algorithm,
                                               sub_batches=sub_batches, timings=timings)
        for t in timings:
            add_timing(acc, t)
        return [m for _, m in parts], sum(len(m) for _, m in parts), acc

    def step_c4():
        """Reference genome and SNP genome, one after the other (the reference runs them as two jobs)."""
        acc = new_acc()
        h_ref = genome.search(codes, max_mm, algorithm=algorithm)
        add_timing(acc, ctx.timing())
        h_snp = snp_genome.search(codes, max_mm, algorithm=algorithm)
        add_timing(acc, ctx.timing())
        return [h_ref, h_snp], len(h_ref) + len(h_snp), acc

    def step_streamed():
        """c5: the reads are streamed in batches; every batch's hits get their packed feature rows (+ MIT score)
        on the GPU that owns the shard, before any gather.  One rank: the whole loop runs behind the C ABI
        (vsc_search_stream, scoring from the batch callback)."""
        acc = new_acc()
        if not use_dist:
            if rows_fused:
                genome.search_streamed_rows(codes, max_mm, lambda h, first, count, rows_dev: None, batch=args.batch, algorithm=algorithm)
            else:
                genome.search_streamed(codes, max_mm, score_batch, batch=args.batch, algorithm=algorithm)
            add_timing(acc, ctx.timing(), score=True)
            return [], acc["hits"], acc
        # several ranks: every batch is scored on the rank that found the hits, then records (+ votes with the classifier)
        # are gathered to rank 0 while the next batch is searched (varscot_amd.dist.sharded_search_stream)
        total = [0]

        def score(h, first, count):
            add_timing(acc, ctx.timing())
            if forest is not None:
                votes = torch.empty(max(len(h), 1), dtype=torch.int16, device=device)
                forest.classify_hits(h, guide_activity[first:first + count], to_host=False, dev_ptr=votes.data_ptr())
                acc["score_ms"] += ctx.timing()["score_ms"]
                v = votes[:len(h)]
                return v if xdev is not None else v.cpu()
            h.packed_features(to_host=False, mit=False)
            acc["score_ms"] += ctx.timing()["score_ms"]
            return None

        def on_batch(mh, first, count, votes):
            if mh is not None:
                total[0] += len(mh)

        vdist.sharded_search_stream(ctx, genome, codes, max_mm, on_batch, args.batch, device=xdev, algorithm=algorithm, score=score)
        return [], total[0], acc

    def step():
        if args.workload == "c4":
            return step_c4()
        if pipelined and mode.get("pipelined"):
            return step_pipelined()
        if streamed:
            return step_streamed()
        acc = new_acc()
        h, m = search_batch(codes, acc)
        return [x for x in (h, m) if x is not None], len(m) if m is not None else len(h), acc

    def timed_run():
        """W untimed steps, then exactly K steps between barriers; max over ranks."""
        for i in range(args.warmup):
            objs, _, _ = step()
            for o in objs:
                o.close()
        barrier()
        for k in X_KEYS:
            xacc[k] = 0.0
        sums, total_hits = new_acc(), 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            objs, total_hits, acc = step()
            add_timing(sums, acc, score=True)
            for o in objs:
                o.close()
        barrier()
        dt = time.perf_counter() - t0
        x = dict(xacc)
        if use_dist:
            tt = torch.tensor([dt] + [x[k] for k in X_KEYS[:4]], dtype=torch.float64, device=xdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bb = torch.tensor([x["sent_bytes"], x["received_bytes"]], dtype=torch.float64, device=xdev)
            dist.all_reduce(bb, op=dist.ReduceOp.SUM)
            dt = float(tt[0].item())
            for i, k in enumerate(X_KEYS[:4]):
                x[k] = float(tt[1 + i].item())
            x["sent_bytes"], x["received_bytes"] = float(bb[0].item()), float(bb[1].item())
        return dt, sums, total_hits, x

    exchanges = None
    if use_dist:
        # both shapes of the one exchange, K steps each (plain: search, pack, exchange, merge, one after the other)
        exchanges = {}
        # (the streamed, scored workload always gathers to rank 0: one run)
        for ex in (("root",) if streamed else ("root", "reads") if args.exchange == "reads" else ("reads", "root")):
            mode.update(exchange=ex, pipelined=False)
            dt_x, sums_x, hits_x, x = timed_run()
            exchanges[ex] = {
                "value": n_guides * args.steps / dt_x, "unit": "guides/s", "ms_per_step": dt_x / args.steps * 1e3,
                "search_wall_ms": x["search_ms"] / args.steps, "pack_ms": x["pack_ms"] / args.steps,
                "exchange_ms": x["exchange_ms"] / args.steps, "merge_ms": x["merge_ms"] / args.steps,
                "exchanged_bytes": x["sent_bytes"] / args.steps,
                "record_bytes": vdist.XREC_BYTES,
                "result": "all records on rank 0" if ex == "root" else "every rank holds the merged records of its read range",
                "note": "host wall times per step, max over ranks; exchanged_bytes = bytes that left a rank, summed over ranks"}
            keep = (dt_x, sums_x, hits_x)
        mode.update(exchange=args.exchange, pipelined=True)
        if pipelined:
            dt, sums, total_hits, _ = timed_run()   # the headline: --exchange's gather with the exchange of one piece
        else:                                       # hidden behind the search of the next
            dt, sums, total_hits = keep             # (the last plain run was --exchange's)
    else:
        dt, sums, total_hits, _ = timed_run()
    hits_local, sites_local, passes = sums["hits"] // args.steps, sums["sites"], sums["passes"]
    pairs_local, stream_bytes = sums["pairs"] // args.steps, sums["genome_bytes"] // args.steps
    if use_dist:
        # every rank holds a part of the result (exchange = reads) or rank 0 holds it all (root): sum of the shard hits
        agg = torch.tensor([float(sites_local), float(hits_local)], dtype=torch.float64, device=xdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        total_sites = float(agg[0].item())
        total_hits = int(agg[1].item())
    else:
        total_sites = float(sites_local)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        guides_per_s = n_guides * args.steps / dt
        # ---- roofline ---------------------------------------------------------------------------------------
        # roofline.frac is the WHOLE STEP's: the 8(d) bytes of the step (what ANY search of this shape has to move: per genome
        # pass 0.375 B/base of planes, + 16 B per hit + 16 B per read [+ 16 + 64 B per hit with per-hit feature rows]) over the
        # step's time and the 8 TB/s peak.  Beside it every kernel of the step on ITS OWN bytes (roofline.kernels) - the
        # bytes its data structure makes it move - and, for the search kernel, the instruction-issue roofline from the
        # committed counters of the same command (profiles/, keyed by a hash of the kernel sources: stale -> null).
        own_bases = (we - wb) * 32
        scan_avg_ms = sums["scan_ms"] / args.steps
        read_passes = -(-n_guides // args.batch) if streamed else max(1, sums["read_passes"])
        scored = streamed and forest is None
        survey_bytes = 0.375 * own_bases * read_passes + 16.0 * hits_local + 16.0 * n_guides + (80.0 * hits_local if scored else 0.0)
        if algorithm == "scan":
            kernel, ops_per_compare = "scan_kernel", LANE_OPS_PER_COMPARE["scan"]
            structure_bytes = 0.375 * own_bases * read_passes + 12.0 * hits_local + 8.0 * n_guides
            structure_model = "planes streamed once per pass (0.375 B/base) + 12 B per hit written + 8 B per read"
        else:
            kernel, ops_per_compare = "seed_sliced_kernel", LANE_OPS_PER_COMPARE["sliced"]
            # read-list entries: as many as the library made (vsc_timing.list_entries; the cut of the pigeonhole - SeedPlan,
            # vsc_internal.h - is its cost model's choice: vsc_timing.seed_cut)
            list_entries = sums["list_entries"] / args.steps
            structure_bytes = float(stream_bytes) + 8.0 * list_entries + 16.0 * hits_local
            structure_model = ("bit-sliced sites of the visited buckets (3.5 B each, read once) + read-list entries (8 B each) + "
                               "per hit one 8 B site record read and one 8 B record written")
        compares = float(pairs_local)
        lane_ops = compares * ops_per_compare / (scan_avg_ms * 1e-3)
        # counters of the same kernel from the PMC passes of tools/collect_profiles.sh (one counter group per run), looked
        # up - not measured in this run - and only if they were taken with the kernels this run executes
        traffic, traffic_source, counted = None, None, None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("_kernel_sources_sha") == kernel_sources_sha():
                    traffic = tj.get("%s/%s/%d" % (args.workload, algorithm, world))
                    counted = tj.get("%s/%s/%d:counters" % (args.workload, algorithm, world))
                    traffic_source = tj.get("_source", "profiles/scan_traffic.json") if traffic is not None else None
                else:
                    traffic_source = ("profiles/scan_traffic.json was collected with other kernel sources (%s) than this run's (%s): "
                                      "not reported" % (tj.get("_kernel_sources_sha"), kernel_sources_sha()))
            except Exception:
                traffic = None
        valu = {"pair_compares_per_s": compares / (scan_avg_ms * 1e-3), "lane_ops_per_compare": ops_per_compare,
                "achieved_lane_ops_per_s": lane_ops, "peak_lane_ops_per_s": VALU_LANE_OPS_PEAK, "frac": lane_ops / VALU_LANE_OPS_PEAK,
                "issue": None}
        if counted and counted.get("simd_cycles"):
            v, sc = counted["SQ_INSTS_VALU"], counted["simd_cycles"]
            valu["issue"] = {
                "valu_wave_instructions": v, "salu_wave_instructions": counted.get("SQ_INSTS_SALU"), "simd_cycles": sc,
                "valu_per_simd_cycle": v / sc, "issue_cycles_per_instruction": VALU_ISSUE_CYCLES,
                # share of the SIMDs' cycles the kernel's vector instructions take at the measured issue costs: between "every
                # instruction has vector operands only" and "every instruction has one scalar operand"
                "frac_low": v * VALU_ISSUE_CYCLES["vector_operands"] / sc, "frac_high": v * VALU_ISSUE_CYCLES["scalar_operand"] / sc,
                "source": "SQ_INSTS_VALU / GRBM_GUI_ACTIVE of profiles/*_seed_pmc.json x the issue costs of profiles/r03_valu_rate_microbench.txt"}

        def kernel_entry(name, ms, nbytes, model, **extra):
            d = {"kernel": name, "ms": ms, "bytes": nbytes, "bytes_model": model,
                 "achieved": (nbytes / (ms * 1e-3) / 1e9) if ms else None, "unit": "GB/s",
                 "frac": (nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None}
            d.update(extra)
            return d

        sort_ms, fin_ms, score_ms = sums["sort_ms"] / args.steps, sums["finalize_ms"] / args.steps, sums["score_ms"] / args.steps
        fin_bytes = (24.0 + (4.0 + 64.0 if rows_fused else 0.0)) * hits_local
        part_bytes = max(0.0, sums["sort_bytes"] / args.steps - (24.0 + (80.0 if rows_fused else 0.0)) * hits_local)
        kernels = [kernel_entry(kernel, scan_avg_ms, structure_bytes, structure_model, valu=valu, hbm_traffic_measured=traffic),
                   kernel_entry("bin_partition_kernel (+ bin_hist / bin_scan where a level needs them)", sort_ms, part_bytes,
                                "8 B read + 8 B written per record and partition level (+ 8 B per record for a histogram pass)"),
                   kernel_entry("bin_finalize_kernel" + ("<rows>" if rows_fused else ""), fin_ms, fin_bytes,
                                "8 B record + 4 B side word read, 16 B vsc_hit + 64 B packed feature row written per hit (record and side word "
                                "a second time from L2)" if rows_fused else "8 B record read + 16 B vsc_hit written per hit")]
        if streamed and not rows_fused:
            extra = {}
            if forest is not None:
                # the forest walk is bound by vector-instruction issue: the committed counters of the kernel (same
                # sources, else null) x the cost of its instruction mix (tools/micro/valu_kinds.hip: 16 instructions, 40 SIMD-cycles)
                fc = None
                try:
                    tj = json.load(open(tpath))
                    if tj.get("_kernel_sources_sha") == kernel_sources_sha():
                        fc = tj.get("c5/forest:counters")
                except Exception:
                    fc = None
                extra["valu"] = {"issue": None if not fc else {
                    "valu_per_simd_cycle": fc["valu_per_simd_cycle"], "cycles_per_instruction_of_the_mix": FOREST_CYCLES_PER_INSTRUCTION,
                    "frac": fc["valu_per_simd_cycle"] * FOREST_CYCLES_PER_INSTRUCTION,
                    "lds_array_cycles_per_cu_cycle": fc["lds_array_cycles_per_cu_cycle"],
                    "source": "SQ_INSTS_VALU, SQ_LDS_IDX_ACTIVE / GRBM_GUI_ACTIVE of profiles/*_seed_pmc.json (forest) x the issue costs of "
                              "profiles/r04_valu_kinds_microbench.txt"}}
            kernels.append(kernel_entry("rf_predict_kernel<fused>" if forest is not None else "score_packed_kernel", score_ms,
                                        (18.0 if forest is not None else 80.0) * hits_local,
                                        "16 B vsc_hit read + 2 B of votes written per hit (the forest walk is bound by vector-instruction issue, "
                                        "not memory: valu.issue; DESIGN.md 4.6)"
                                        if forest is not None else "16 B vsc_hit read + 64 B packed row written per hit", **extra))
        whole = survey_bytes / (ms_per_step * 1e-3) / 1e9
        dominant = max(kernels, key=lambda k: k["ms"] or 0.0)
        out = {
            "metric": "guides/sec (candidate sites/sec alongside) at <=%d mismatches on a %.1f Gbp reference" % (max_mm, total_bases / 1e9),
            "value": guides_per_s, "unit": "guides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "guides": n_guides, "genome_bases": total_bases,
                       "max_mismatches": max_mm, "parallelism": "genome-shard x%d" % world, "algorithm": algorithm,
                       "multi_gpu_path": ("torch.distributed (RCCL) one process per GPU: varscot_amd/dist.py" if use_dist else None),
                       "exchange": (args.exchange if use_dist else None), "sub_batches": (sub_batches if pipelined else 1),
                       "hooks": (hooks or None),
                       "rccl_ranks": (dist.get_world_size() if use_dist else None),
                       "backend": (dist.get_backend() if use_dist else None),
                       "hits_per_step": int(total_hits), "candidate_sites_per_s": total_hits * args.steps / dt,
                       "pam_valid_sites": int(total_sites), "search_launches": passes, "read_passes": read_passes,
                       "batch": args.batch if streamed else n_guides, "variant_genome": snp_info,
                       "per_hit_scoring": (None if not streamed else
                                           "score -> classify fused: rfClassifier (1000 trees) walked per hit, 2 B of votes per hit"
                                           if forest is not None else "64-byte packed feature rows (442 features) per hit, written by the record "
                                           "assembly (vsc_search_stream_rows)" if rows_fused else "64-byte packed feature rows (442 features) per hit, "
                                           "second pass (vsc_score_hits_packed)")},
            "roofline": {"bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": whole / HBM_PEAK_GBS,
                         "scope": "whole step: SURVEY.md 8(d) bytes of the step / ms_per_step / 8 TB/s",
                         "algorithmic_bytes": survey_bytes,
                         "bytes_model": "SURVEY.md 8(d): 0.375 B/base per genome pass + 16 B/hit + 16 B/read"
                                        + (" + (16 + 64) B/hit for the per-hit feature rows" if scored else ""),
                         "kernel": dominant["kernel"], "launch_ms": dominant["ms"],
                         "traffic": traffic if dominant["kernel"] == kernel else None, "traffic_source": traffic_source,
                         "kernels": kernels,
                         "note": "the search kernel is an integer/bitwise compare kernel bound by vector-instruction issue, not by "
                                 "HBM (DESIGN.md section 4.2): its entry carries both rooflines; the sort kernels are the HBM-bound ones"},
            # the ordering of the hits (hand-written bin sort) is the second largest share of a step at m = 8
            "roofline_sort": sort_roofline(sums, args.steps),
            "kernels_ms": {"search": scan_avg_ms, "prep": sums["prep_ms"] / args.steps, "sort": sums["sort_ms"] / args.steps,
                           "finalize": sums["finalize_ms"] / args.steps,
                           "score": (sums["score_ms"] / args.steps) if streamed else None},
            "seed_cut": (None if algorithm == "scan" else
                         {"segment_0_within": sums["seed_cut"] & 15, "segment_1_within": sums["seed_cut"] >> 4,
                          "segment_2_within": "what the site's PAM leaves of the limit - %d - %d - 2" % (sums["seed_cut"] & 15, sums["seed_cut"] >> 4),
                          "list_entries_per_step": sums["list_entries"] / args.steps, "chosen_by": "the library's cost model (vsc_api.cpp)"}),
            "setup": {"genome_generate_s": t_gen, "genome_hbm_bytes": genome.device_bytes, "index_build_ms": index_ms},
        }
        if exchanges is not None:
            out["exchanges"] = exchanges
            head = exchanges.get(args.exchange) or exchanges["root"]
            out["exchange_ms"], out["exchanged_bytes"] = head["exchange_ms"], head["exchanged_bytes"]
        if use_dist:
            out["n_gt_1_rccl_executed"] = bool(world > 1 and dist.get_backend() == "nccl")
        if abi_line is not None:
            # the same workload through the product's own multi-device driver (one process over the devices, vsc_multi_*)
            out["multi_abi"] = abi_line if "error" in abi_line else {
                k: abi_line.get(k) for k in ("value", "unit", "ms_per_step", "n_gpus", "steps", "config", "vsc_multi_timing", "n_gt_1_rccl_executed")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(total_bases, max_mm, args.cpu_sample_bases)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    genome.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
