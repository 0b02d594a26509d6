"""bench.py's stdout contract, checked WITHOUT a GPU: the compact line built from a full record of realistic content (the last
committed full record under profiles/) stays under the size the driver's parser keeps, carries every contract key, and refers to
the offline records by file + sha instead of embedding them (a 20.3 KB line left BENCH_r05.parsed null)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (module import only: no torch, no GPU at import time)


def _latest_full_record():
    """A full record: the r06 form if one is committed, else the r05 line (which WAS the full record) with its embedded bodies
    replaced by the references bench.py now forms."""
    new = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_full.json")))
    if new:
        return json.load(open(new[-1]))
    d = json.load(open(os.path.join(ROOT, "profiles", "r06z_bench.json")))
    _, ref = bench.load_record(bench.TRAFFIC_FILE)
    d["roofline"]["traffic_source"] = ref
    d["hommul_relin_rescale"]["stages"] = bench.stage_ref(bench.STAGES_FILE)
    d["hommul_relin_rescale"]["batched"]["stages"] = bench.stage_ref(bench.STAGES_BATCHED_FILE)
    return d


def test_compact_line_fits_and_keeps_the_contract():
    full = _latest_full_record()
    text = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(text) <= bench.LINE_LIMIT == 8000 and "\n" not in text
    assert len(text) <= 5000, len(text)            # margin: the r04 line (11.6 KB) parsed, the r05 line (20.3 KB) did not
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "sustained"):
        assert r.get(k) is not None, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c)
    assert d["full_record"] == "gpurun_out/bench_full.json"


def test_offline_records_are_referenced_not_embedded():
    body, ref = bench.load_record(bench.TRAFFIC_FILE)
    assert set(ref) == {"file", "sha16", "collected"} and body["ntt_batched_bytes_per_launch"] > 0
    for path in (bench.STAGES_FILE, bench.STAGES_BATCHED_FILE):
        ref = bench.stage_ref(path)
        assert "stages" not in ref and "batches" not in ref and len(json.dumps(ref)) < 500
        assert 0.0 < ref["frac"] < 1.0 and ref["furthest_below_roofline"]


def test_an_oversized_record_sheds_optional_legs_instead_of_breaking_the_line():
    full = _latest_full_record()
    full["keyswitch_c4"] = dict(full["keyswitch_c4"] or {}, checked="x" * 9000)
    d = json.loads(bench.compact_line(full, "f.json"))
    assert isinstance(d["keyswitch_c4"], str) and d["keyswitch_c4"].startswith("dropped") and "roofline" in d and "cpu_baseline" in d
