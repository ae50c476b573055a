"""bench.py's bookkeeping, on CPU: the algorithmic-bytes formula of SURVEY.md 8(d) over the committed work counts, and the
lookup of the committed counter passes (profiles/k_trace_traffic.json) by scene, overrides, image size, launch shape and device code."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (no torch / GPU at import)


def test_algorithmic_bytes_follow_the_survey_formula():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "workcounts.json")))
    c1 = bench.committed_workcounts("sphere-hairblock", {}, 720, 1.0)
    assert c1 is fx["C1"] or c1 == fx["C1"]
    p = c1["per_sample"]
    want = 32 * p["nodes"] + 44 * p["seg_tests"] + 52 * p["tri_tests"] + 104 * p["hair_shades"] + 88 * p["env_samples"] + 32.0 / 64
    assert abs(bench.algorithmic_bytes_per_sample(p, 64, env_textured=False) - want) < 1e-9
    assert abs(want - c1["algorithmic_bytes_per_sample"]) < 0.6 and not c1["env_textured"]  # a constant environment reads no texels
    # a textured environment adds 48 B per lookup
    assert abs(bench.algorithmic_bytes_per_sample(p, 64, env_textured=True) - want - 48 * p["env_lookups"]) < 1e-9
    # scene overrides select the fixture: C2 at the middle of its beta_m sweep, and nothing for an override that was never counted
    assert bench.committed_workcounts("straight-hair", {"beta_m": 0.25}, 720, 1.0) == fx["C2-beta_m0.25"]
    assert bench.committed_workcounts("straight-hair", {"beta_m": 0.4}, 720, 1.0) is None
    assert bench.committed_workcounts("curly-hair", {}, 1280, 0.5) is None  # another geometry


def test_counter_passes_are_matched_by_overrides_shape_and_device_code(monkeypatch):
    allp = json.load(open(os.path.join(ROOT, "profiles", "k_trace_traffic.json")))
    c2 = [e for e in allp if e["scene"] == "straight-hair"]
    assert c2 and c2[0].get("scene_kw") == {"beta_m": 0.25}  # C2's pass was taken on the config the driver's line runs
    e = c2[0]
    monkeypatch.setattr(bench, "csrc_sha16", lambda: e["csrc_sha16"])
    hit, why = bench.committed_counters("straight-hair", {"beta_m": 0.25}, e["resolution"], 1.0, 1, e["launch_shape"])
    assert hit == e and why is None
    assert bench.committed_counters("straight-hair", {}, e["resolution"], 1.0, 1, e["launch_shape"])[0] is None           # other overrides
    assert bench.committed_counters("straight-hair", {"beta_m": 0.25}, e["resolution"], 1.0, 1, 1)[0] is None              # another kernel
    assert bench.committed_counters("straight-hair", {"beta_m": 0.25}, e["resolution"], 1.0, 2, e["launch_shape"])[0] is None  # N > 1
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "0" * 16)
    hit, why = bench.committed_counters("straight-hair", {"beta_m": 0.25}, e["resolution"], 1.0, 1, e["launch_shape"])
    assert hit is None and why.startswith("stale")  # taken on other device code: not a statement about this run
    assert bench.valu_block(None) is None and bench.valu_block(e)["bound"] == "valu"


def test_committed_counter_passes_describe_the_committed_device_code():
    """The passes quoted by the driver's line must have been taken on the kernels in the tree (else every line says 'stale')."""
    allp = json.load(open(os.path.join(ROOT, "profiles", "k_trace_traffic.json")))
    assert {(e["scene"], e["launch_shape"]) for e in allp} >= {("sphere-hairblock", 5), ("straight-hair", 3), ("curly-hair", 3), ("hair-curls", 1)}
    stale = [e["config"] for e in allp if e["csrc_sha16"] != bench.csrc_sha16()]
    if stale:  # visible, not fatal: the device code was edited after the passes — bench.py then reports `traffic: null, traffic_source: "stale: ..."`
        import pytest
        pytest.xfail(f"counter passes taken on other device code: {stale}; retake with tools/profile_configs.sh + tools/traffic_from_pmc.py")
