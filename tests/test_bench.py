"""bench.py's launch path: `python bench.py --gpus N` started bare (no launcher environment) brings up the N ranks
itself - torch.distributed.run as a child process, before anything touches a GPU - and relays rank 0's JSON line.
CPU: the gloo dry run (process group + one all-reduce).  GPU box: the whole multi-rank step on the one GPU
(--rehearse: gloo, records through host memory) with both exchanges reported."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout,
                       env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n", [2, 3])
def test_bare_start_brings_up_the_ranks(n):
    r, line = _bench("--gpus", str(n), "--rehearse", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line == {"dry_run": True, "n_gpus": n, "rccl_ranks": n, "backend": "gloo", "ranks_seen": n}
    assert r.stdout.count("{") == 1  # ONE JSON line on stdout


def test_a_launchers_environment_is_respected():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py does not start ranks of its own; a wrong --gpus is an error."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_report_both_exchanges():
    r, line = _bench("--gpus", "2", "--rehearse", "--workload", "c1", "--mismatches", "8", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "gloo"
    ex = line["exchanges"]
    assert set(ex) == {"root", "reads"}
    for k in ("root", "reads"):
        assert ex[k]["value"] > 0 and ex[k]["exchange_ms"] >= 0 and ex[k]["record_bytes"] == 8
    assert ex["root"]["exchanged_bytes"] > 0 and line["exchange_ms"] == ex[line["config"]["exchange"]]["exchange_ms"]
    assert line["config"]["hits_per_step"] > 0
    # the headline is the north star's single gather; the product's own multi-device driver (one process over the devices
    # behind the C ABI) ran the same workload first, as a child of rank 0, and its line is embedded
    # (in four pieces, a piece travelling while the next is searched; the plain one-piece runs are the `exchanges` entries)
    assert line["config"]["exchange"] == "root" and line["config"]["sub_batches"] == 4 and line["value"] > 0
    abi = line["multi_abi"]
    assert "error" not in abi, abi
    assert abi["n_gpus"] == 2 and abi["config"]["devices"] == [0, 0] and abi["value"] > 0
    assert abi["config"]["hits_per_step"] == line["config"]["hits_per_step"]
    assert abi["vsc_multi_timing"]["total_ms"] > 0 and abi["n_gt_1_rccl_executed"] is False and line["n_gt_1_rccl_executed"] is False


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--classify"]])
def test_one_process_over_several_devices_behind_the_abi(extra):
    """`bench.py --multi abi`: vsc_multi_search (c1) and vsc_multi_search_stream with the scoring on the owning shard (a small c5)
    over three contexts on the one GPU - same line as the other driver, the library's own phase times."""
    r, line = _bench("--multi", "abi", "--gpus", "3", "--abi-devices", "0,0,0", "--workload", "c1", "--mismatches", "6", "--steps", "2",
                     "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 3 and line["config"]["exchange"] == "root" and line["config"]["rccl_ranks"] == 0
    assert "device copies" in line["config"]["exchange_transport"] and line["config"]["hits_per_step"] > 0
    t = line["vsc_multi_timing"]
    assert t["search_wall_ms"] > 0 and t["merge_ms"] > 0 and t["total_ms"] >= t["search_wall_ms"]
    r, line5 = _bench("--multi", "abi", "--gpus", "2", "--abi-devices", "0,0", "--workload", "c5", "--guides", "40", "--bases", "2000000",
                      "--batch", "16", "--steps", "1", "--warmup", "1", *extra)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line5["config"]["batches"] == 3 and line5["config"]["hits_per_step"] > 0
    assert (line5["vsc_multi_timing"]["score_ms_max"] > 0) == bool(extra)  # (feature rows: written by the record assembly, no kernel of their own)
    assert ("classify" in line5["config"]["per_hit_scoring"]) == bool(extra)


@pytest.mark.gpu
def test_one_gpu_line_has_the_contract_fields():
    r, line = _bench("--workload", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line
    assert line["n_gpus"] == 1 and "exchanges" not in line and "multi_abi" not in line
    # roofline.frac is the whole step's; every kernel of the step is listed on its own bytes
    roof = line["roofline"]
    assert abs(roof["frac"] - roof["algorithmic_bytes"] / (line["ms_per_step"] * 1e-3) / 1e9 / roof["peak"]) < 1e-9
    names = [k["kernel"] for k in roof["kernels"]]
    assert names[0] in ("seed_sliced_kernel", "scan_kernel") and any("partition" in n for n in names) and "bin_finalize_kernel" in names
    assert all(k["bytes"] >= 0 and k["ms"] >= 0 for k in roof["kernels"])
    assert "issue" in roof["kernels"][0]["valu"]
