"""bench.py contract on a GPU box: one JSON line with the fields the driver reads, and the multi-rank path
(`--gpus 2` with no launcher: bench.py starts its own ranks) reproducing the one-rank results."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra_args, **extra_env):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):   # never inherit a launcher's rendezvous
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    return json.loads(lines[0])


def test_bench_line_contract(gpu):
    d = _run(["--steps", "20", "--warmup", "3", "--no-cpu-baseline"], PHA_BENCH_BATCH="2")
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
    assert d["hommul_relin_rescale"]["value"] > 0 and d["hommul_relin_rescale"]["batched"]["batch"] == 2
    assert d["keyswitch_c4"]["batch"] == 64 and d["keyswitch_c4"]["value"] > 0


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


def test_bench_one_rank_through_rccl(gpu):
    """PHA_BENCH_FORCE_DIST=1: one rank with an RCCL process group (backend "nccl"), so that the key broadcast, the
    max-over-ranks all-reduce, the barriers and the checksum all-gather of the multi-GPU path really execute RCCL
    collectives on this one-GPU box; results equal the plain one-rank run."""
    plain = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1")
    rccl = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"], PHA_BENCH_SMALL="1", PHA_BENCH_FORCE_DIST="1")
    assert rccl["collectives"].startswith("RCCL") and plain["collectives"].startswith("none")
    assert rccl["keyswitch_c4"]["checksum"] == plain["keyswitch_c4"]["checksum"] and rccl["n_gpus"] == 1
