"""bench.py contract on a GPU box: one JSON line with the fields the driver reads, and the multi-rank path
(`--gpus 2` with no launcher: bench.py starts its own ranks) reproducing the one-rank results."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


LINE_LIMIT = 8000     # bytes; bench.py asserts the same (a 20 KB line left the driver's record unparsed)


def _run(extra_args, full=True, **extra_env):
    """The FULL record of a bench run (what r01-r05 printed), after checking the stdout contract: exactly one JSON line, at most
    LINE_LIMIT bytes, naming the file the full record went to.  full=False returns the compact line itself."""
    import tempfile
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):   # never inherit a launcher's rendezvous
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "full.json")
        pre = "--preflight" in extra_args
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args + ([] if pre else ["--full-out", path]),
                             capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout + out.stderr
        line = json.loads(lines[0])
        if pre:
            return line
        assert len(lines[0]) <= LINE_LIMIT, len(lines[0])
        assert line["full_record"] == path
        record = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):   # same run, same numbers
        assert line[k] == record[k] or abs(line[k] - record[k]) <= 1e-5 * abs(record[k]), k
    return record if full else line


def test_bench_compact_line(gpu):
    """The line the driver parses: every contract key, `roofline` and `cpu_baseline` with their meaning intact, references (file +
    sha) instead of the bodies of the offline records, and a size that fits the driver's parser several times over."""
    d = _run(["--steps", "20", "--warmup", "5", "--sustain", "1.0"], full=False, PHA_BENCH_BATCHES="1,8")
    assert len(json.dumps(d, separators=(",", ":"))) <= LINE_LIMIT
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "u64" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "sustained", "traffic_source"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert set(r["traffic_source"]) == {"file", "sha16", "collected"} and r["traffic_source"]["file"] == "profiles/traffic.json"
    assert r["traffic"] > r["algorithmic_bytes_per_launch"] and 1.0 < r["traffic_ratio"] < 4.0
    assert abs(r["sustained"]["median_ms_per_step"] - d["ms_per_step"]) / d["ms_per_step"] < 0.25
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 100 and c["unit"] == "NTT/s" and c["sample"] and c["all_cores"] >= 1
    hm = d["hommul_relin_rescale"]
    assert hm["value"] > 0 and set(hm["batched"]["ms_per_op_by_batch"]) == {"1", "8"}
    for ref in (hm["stages"], hm["batched"]["stages"]):     # references, never bodies
        assert set(ref) <= {"file", "sha16", "collected", "batch", "per_op_us", "furthest_below_roofline", "frac"} and 0 < ref["frac"] < 1
    assert d["keyswitch_c4"]["checked"].startswith("2 ciphertexts == oracle") and d["matvec_c5"]["value"] > 0
    assert d["next_rows"]["modular_gemm"]["frac_of_i8_mfma_peak"] > 0.05


def test_bench_line_contract(gpu):
    d = _run(["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--sustain", "1.5"], PHA_BENCH_BATCHES="1,2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "u64" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.01 < r["frac"] < 1.0
    assert 0.2 < r["ceiling_two_pass"] < 0.6 and abs(r["frac_of_ceiling"] - r["frac"] / r["ceiling_two_pass"]) < 1e-9
    # value and the per-step time describe the same measurement: polynomials_per_step x 45 limb-transforms per step
    per_step = d["config"]["polynomials_per_step"] * 45
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the working set of a step is larger than the MALL
    assert d["config"]["polynomials_per_step"] * 45 * 65536 * 8 > 256 << 20
    assert r["per_step_events"]["steps"] >= 100 and r["per_step_events"]["median_ms"] > 0
    assert d["single_polynomial"]["mall_resident"]["value"] > 0 and d["single_polynomial"]["hbm_resident"]["value"] > 0
    hb = d["hommul_relin_rescale"]["batched"]
    assert d["hommul_relin_rescale"]["value"] > 0 and [e["batch"] for e in hb["sweep"]] == [1, 2] and hb["batch"] in (1, 2)
    assert abs(hb["ms_per_op"] - min(e["ms_per_op"] for e in hb["sweep"])) < 1e-9          # the quoted batch is the sweep's best
    assert d["keyswitch_c4"]["batch"] == 64 and d["keyswitch_c4"]["value"] > 0
    assert d["keyswitch_c4"]["checked"] is None                                            # --no-cpu-baseline: the oracle is not touched
    # r04 calibration: read-only / write-only / copy / in-place streams, each at a plausible fraction of the 8 TB/s line
    for k in ("read_GBps", "write_GBps", "copy_GBps", "rmw_GBps"):
        assert 2000.0 < r[k] < 8000.0, (k, r[k])
    assert abs(r["ceiling_two_pass"] - r["rmw_GBps"] / 2 / 8000.0) < 1e-9
    # r05: the sustained legs (the same step / the best batch back to back for seconds, medians over event-bracketed chunks)
    su = r["sustained"]
    assert su["seconds"] >= 1.5 and su["steps"] >= 100 and su["min_ms_per_step"] <= su["median_ms_per_step"] <= su["max_ms_per_step"]
    assert 0.5 < su["gpu_busy_fraction"] <= 1.001 and abs(su["value"] - per_step / (su["median_ms_per_step"] * 1e-3)) / su["value"] < 1e-6
    hs = hb["sustained"]
    assert hs["batch"] == hb["batch"] and hs["seconds"] >= 0.75 and hs["median_ms_per_op"] > 0
    assert hb["fixed_batch_8"] is None            # B = 8 is not in this run's sweep
    assert "kernel_memory_floor_ms" not in r and r["kernel_memory_floor_ms_offline"] > 0   # an offline constant is named as one
    # r06: offline records are referenced, not embedded
    assert set(r["traffic_source"]) == {"file", "sha16", "collected"}
    assert "stages" not in (d["hommul_relin_rescale"]["stages"] or {}) and "batches" not in (hb["stages"] or {})


def test_bench_preflight(gpu):
    """`bench.py --gpus N --preflight`: one JSON line, exit code 0, the trial broadcast through BOTH paths on a
    one-rank RCCL group; without a group it says so; two ranks on one device (gloo) pass the dist.broadcast path and skip the other."""
    one = _run(["--preflight"])
    assert one["preflight"] is True and one["ok"] is True and one["n_gpus"] == 1 and one["notes"]
    rccl = _run(["--preflight"], PHA_BENCH_FORCE_DIST="1")
    assert rccl["ok"] is True and rccl["backend"] == "nccl" and rccl["errors"] == []
    tb = rccl["per_rank"][0]["trial_broadcast"]
    assert tb["dist.broadcast"]["ok"] and tb["dist.broadcast"]["path_taken"] == "dist.broadcast"
    assert tb["pha_broadcast_keys"]["ok"] and tb["pha_broadcast_keys"]["path_taken"] == "pha_broadcast_keys"
    assert rccl["per_rank"][0]["free_GB"] >= rccl["need_GB_per_rank"]
    two = _run(["--gpus", "2", "--preflight"], PHA_BENCH_SHARE_GPU="1")
    assert two["ok"] is True and two["n_gpus"] == 2 and [e["rank"] for e in two["per_rank"]] == [0, 1]
    assert all(e["trial_broadcast"]["dist.broadcast"]["ok"] for e in two["per_rank"])
    assert all("skipped" in e["trial_broadcast"]["pha_broadcast_keys"] for e in two["per_rank"])


def test_bench_two_ranks_self_spawned_reproduce_one_rank(gpu):
    """`python bench.py --gpus 2` starts its own two ranks (here both on cuda:0 over gloo: the box has one GPU and
    RCCL refuses two ranks on one device); keys go through the broadcast, the config-4 batch is split by shard_range,
    and the checksum over all output ciphertexts equals the one-rank run's."""
    one = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1")
    two = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"],
               PHA_BENCH_SMALL="1", PHA_BENCH_SHARE_GPU="1")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["keyswitch_c4"]["per_rank_ciphertexts"] == [3, 3] and one["keyswitch_c4"]["per_rank_ciphertexts"] == [6]
    assert one["keyswitch_c4"]["checksum"] == two["keyswitch_c4"]["checksum"]
    assert two["value"] > 0 and two["hommul_relin_rescale"]["value"] > 0
    # the key sets travel as one flat buffer each (one slab per set), not as one collective per [2][#QP][N] tensor
    assert two["key_broadcast_calls"] == {"evk_c3": 1, "keys_c4": 1} and two["matvec_c5"]["key_broadcast_calls"] == 1
    assert one["key_broadcast_calls"] is None
    per_rank = two["rccl"]["per_rank"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    import torch
    if torch.cuda.device_count() >= 2:      # a multi-GPU box (ranks on their own devices): one PCI bus id per rank
        assert len({p["pci_bus_id"] for p in per_rank}) == 2


def test_bench_two_ranks_keep_the_cpu_baseline_and_check_config4(gpu):
    """With the CPU baseline on (the default) a multi-rank line still carries `cpu_baseline` (rank 0 times it while the others
    wait at the barrier behind it) and the config-4 leg states how many ciphertexts it compared with the oracle."""
    two = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-graph", "--no-c5"], PHA_BENCH_SMALL="1", PHA_BENCH_SHARE_GPU="1")
    assert two["n_gpus"] == 2 and two["cpu_baseline"]["value"] > 0 and two["cpu_baseline"]["kind"] == "port"
    assert two["cpu_baseline"]["checked"].startswith("GPU forward NTT")
    assert two["keyswitch_c4"]["checked"].startswith("4 ciphertexts == oracle")      # first and last of each rank's 3


def test_bench_one_rank_through_rccl(gpu):
    """PHA_BENCH_FORCE_DIST=1: one rank with an RCCL process group (backend "nccl"), so that the key broadcast, the
    max-over-ranks all-reduce, the barriers and the checksum all-gather of the multi-GPU path really execute RCCL
    collectives on this one-GPU box; results equal the plain one-rank run."""
    plain = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1")
    rccl = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1", PHA_BENCH_FORCE_DIST="1")
    assert rccl["collectives"].startswith("RCCL") and plain["collectives"].startswith("none")
    assert rccl["keyswitch_c4"]["checksum"] == plain["keyswitch_c4"]["checksum"] and rccl["n_gpus"] == 1
    assert rccl["key_broadcast_calls"] == {"evk_c3": 1, "keys_c4": 1}
    # the opt-in direct form: pha_broadcast_keys (one ncclGroupStart / End around the set) on the process group's OWN communicator
    direct = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1", PHA_BENCH_FORCE_DIST="1",
                  PHA_BCAST_DIRECT="1")
    assert direct["key_broadcast_path"] == "pha_broadcast_keys" and rccl["key_broadcast_path"] == "dist.broadcast"
    assert direct["keyswitch_c4"]["checksum"] == plain["keyswitch_c4"]["checksum"]
    assert direct["matvec_c5"]["checksum"] == plain["matvec_c5"]["checksum"]
