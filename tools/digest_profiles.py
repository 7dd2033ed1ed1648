#!/usr/bin/env python3
"""digest_profiles.py TAG - turn gpurun_out/TAG (tools/collect_profiles.sh) into the committed evidence
under profiles/: bench lines, kernel statistics, one PMC summary, and scan_traffic.json (the measured HBM
bytes per launch that bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402  (what the counters are keyed by: bench.py prints null for other kernels)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def counters(path_glob, kernel_substr):
    """mean per dispatch of every counter over the dispatches of one kernel (warm-up launch included)"""
    acc = {}
    for path in glob.glob(path_glob, recursive=True):
        per = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_substr not in row["Kernel_Name"]:
                    continue
                per.setdefault((row["Counter_Name"], row["Dispatch_Id"]), 0.0)
                per[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        for (name, _), v in per.items():
            acc.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def kernel_avg_ms(stats_csv, kernel_substr):
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Name"]:
                return float(row["AverageNs"]) * 1e-6
    return None


summary = {"round": tag, "kernel": "vsc::seed_sliced_kernel",
           "note": "one rocprofv3 --pmc pass per counter group over `python3 bench.py --workload <w> --steps 3 --warmup 1 "
                   "--no-cpu-baseline` (tools/collect_profiles.sh); FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts "
                   "64 B per 128 B request on gfx950 (MI355X_MICROARCH.md, HBM) and is doubled; averages per launch"}
traffic_path = os.path.join(dst, "scan_traffic.json")
traffic = {}  # only what THIS collection measured: the file is keyed by the hash of the kernel sources it was taken with
for w in ("c2", "c3"):
    bench = json.loads(open(os.path.join(src, "bench_%s.json" % w)).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(dst, "%s_bench_%s_seed.json" % (tag, w)), "w"))
    stats = glob.glob(os.path.join(src, "stats_%s" % w, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_%s_seed_kernel_stats.csv" % (tag, w)))
    c = {}
    for grp in ("sq", "fetch", "write"):
        c.update(counters(os.path.join(src, "pmc_%s_%s" % (grp, w), "**", "*counter_collection.csv"), "seed_sliced_kernel"))
    fetch_raw = c.get("FETCH_SIZE", 0.0) * 1024.0
    write = c.get("WRITE_SIZE", 0.0) * 1024.0
    hbm = 2.0 * fetch_raw + write
    search_entry = bench["roofline"]["kernels"][0]  # the search kernel: first of the per-kernel entries
    launch_ms = search_entry["ms"]
    # GRBM_GUI_ACTIVE is reported once per XCD (8 rows per dispatch, summed above): busy cycles = sum / 8
    busy = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    simd_cycles = busy * 256 * 4
    summary[w] = {
        "counters": c,
        "derived": {"fetch_bytes_raw": fetch_raw, "fetch_bytes_x2": 2.0 * fetch_raw, "write_bytes": write,
                    "hbm_traffic_bytes_per_launch": hbm,
                    "effective_clock_GHz": busy / (launch_ms * 1e-3) / 1e9,
                    "valu_wave_instr_per_simd_cycle": c.get("SQ_INSTS_VALU", 0.0) / simd_cycles if simd_cycles else None,
                    "salu_per_valu": c.get("SQ_INSTS_SALU", 0.0) / max(c.get("SQ_INSTS_VALU", 1.0), 1.0),
                    "wait_any_share_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)},
        "algorithmic_bytes": bench["roofline"]["algorithmic_bytes"], "launch_ms": launch_ms,
        "rocprof_avg_ms": kernel_avg_ms(stats[0], "seed_sliced_kernel") if stats else None,
        "pairs": bench["config"].get("hits_per_step") and search_entry["valu"]["pair_compares_per_s"] * launch_ms * 1e-3,
    }
    traffic["%s/seed/1" % w] = hbm
    traffic["%s/seed/1:counters" % w] = {"SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"),
                                         "simd_cycles": simd_cycles, "fetch_bytes_raw": fetch_raw, "write_bytes": write}
    # the sort kernels of the same runs (round 2 on: vsc_sort.hip)
    sort = {}
    for kname in ("bin_hist_kernel", "bin_partition_kernel", "bin_finalize_kernel"):
        ck = {}
        for grp in ("sq", "fetch", "write"):
            ck.update(counters(os.path.join(src, "pmc_%s_%s" % (grp, w), "**", "*counter_collection.csv"), kname))
        if ck:
            sort[kname] = {"counters": ck, "fetch_bytes_x2": 2048.0 * ck.get("FETCH_SIZE", 0.0), "write_bytes": 1024.0 * ck.get("WRITE_SIZE", 0.0),
                           "rocprof_avg_ms": kernel_avg_ms(stats[0], kname) if stats else None}
    summary[w]["sort_kernels"] = sort
# the kernels that write the per-hit feature rows of c5 (two batches of 10 000 reads), both routes
rows = {}
for route in ("fused", "two-pass"):
    per = {}
    for kname in ("seed_sliced_kernel", "bin_partition_kernel", "bin_finalize_kernel", "score_packed_kernel"):
        ck = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            ck.update(counters(os.path.join(src, "pmc_c5_%s_%s" % (route, ctr), "**", "*counter_collection.csv"), kname))
        stats = glob.glob(os.path.join(src, "stats_c5_%s" % route, "**", "*kernel_stats.csv"), recursive=True)
        if ck:
            per[kname] = {"fetch_bytes_raw_per_launch": 1024.0 * ck.get("FETCH_SIZE", 0.0), "write_bytes_per_launch": 1024.0 * ck.get("WRITE_SIZE", 0.0),
                          "rocprof_avg_ms": kernel_avg_ms(stats[0], kname) if stats else None}
    if per:
        rows[route] = per
    stats = glob.glob(os.path.join(src, "stats_c5_%s" % route, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_c5_%s_kernel_stats.csv" % (tag, route.replace("-", "_"))))
summary["c5_rows"] = dict(rows, note="bench.py --workload c5 --guides 20000 --rows fused | two-pass: per launch = per batch of 10 000 reads (8.2e8 ... "
                                     "1.63e9 hits); FETCH_SIZE raw (the guide's x2 not applied), averages over the launches of the run incl. warm-up")
# the forest walk: pair nodes and the compact nodes of round 3 (one c5 batch of 10 000 reads each + a warm-up batch)
forest = {}
for form in ("pair", "compact"):
    ck = {}
    for grp in ("a", "b"):
        ck.update(counters(os.path.join(src, "pmc_forest_%s_%s" % (form, grp), "**", "*counter_collection.csv"), "rf_predict_kernel"))
    stats = glob.glob(os.path.join(src, "stats_forest_%s" % form, "**", "*kernel_stats.csv"), recursive=True)
    if not ck:
        continue
    ms = kernel_avg_ms(stats[0], "rf_predict_kernel") if stats else None
    busy = ck.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    forest[form] = {"counters": ck, "rocprof_avg_ms": ms,
                    "derived": {"valu_wave_instr_per_simd_cycle": ck.get("SQ_INSTS_VALU", 0.0) / (busy * 1024) if busy else None,
                                "lds_wave_instr_per_cu_cycle": ck.get("SQ_INSTS_LDS", 0.0) / (busy * 256) if busy else None,
                                "lds_array_cycles_per_cu_cycle": ck.get("SQ_LDS_IDX_ACTIVE", 0.0) / (busy * 256) if busy else None,
                                "bank_conflict_share_of_lds_cycles": ck.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(ck.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
                                "valu_per_lds_instr": ck.get("SQ_INSTS_VALU", 0.0) / max(ck.get("SQ_INSTS_LDS", 1.0), 1.0)}}
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_forest_%s_kernel_stats.csv" % (tag, form)))
if forest:
    summary["forest"] = dict(forest, note="bench.py --workload c5 --classify --guides 10000 [--hook rf_form=1]: rf_predict_kernel<2> per launch = one batch "
                                          "(1.63e9 rows x 1 000 trees); SQ_* summed over the chip, GRBM_GUI_ACTIVE / 8 = busy cycles")
summary["kernel_sources_sha"] = kernel_sources_sha()
json.dump(summary, open(os.path.join(dst, "%s_seed_pmc.json" % tag), "w"), indent=1)
for name in ("bench_c4", "bench_c5", "bench_c5_two_pass", "bench_c5_classify", "bench_abi_c3_x1", "bench_abi_c3_x4_one_gpu", "bench_abi_c5_x2_one_gpu", "bench_default"):
    path = os.path.join(src, name + ".json")
    if os.path.exists(path) and os.path.getsize(path):
        shutil.copy(path, os.path.join(dst, "%s_%s.json" % (tag, name)))
if forest.get("pair"):
    fp = forest["pair"]
    traffic["c5/forest:counters"] = {"valu_per_simd_cycle": fp["derived"]["valu_wave_instr_per_simd_cycle"],
                                     "lds_array_cycles_per_cu_cycle": fp["derived"]["lds_array_cycles_per_cu_cycle"],
                                     "valu_per_lds_instr": fp["derived"]["valu_per_lds_instr"], "rocprof_avg_ms": fp["rocprof_avg_ms"]}
traffic["_kernel_sources_sha"] = kernel_sources_sha()
traffic["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh, digested into profiles/%s_seed_pmc.json (FETCH_SIZE x 2 + WRITE_SIZE per launch of the search kernel)" % tag
json.dump(traffic, open(traffic_path, "w"))
print(json.dumps({w: summary[w]["derived"] for w in ("c2", "c3")}, indent=1))
print(json.dumps({w: {k: (v["rocprof_avg_ms"], v["fetch_bytes_x2"], v["write_bytes"]) for k, v in summary[w]["sort_kernels"].items()} for w in ("c2", "c3")}, indent=1))
