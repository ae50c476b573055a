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


def test_the_lines_scalars_and_summary_carry_every_config():
    """the driver's record keeps the scalars of `config` and the tail of the line. bench.summary_scalars on the shape of a real
    line (profiles/r06/bench_C1_driver_flags.json): every config's value / fraction / parity ratio, the 8-GPU projections and the whole command's
    wall clock come out as scalars; a failed leg is an `_error` string and nothing else of it; a leg that did not run leaves no key; and in the
    committed line `summary` is the LAST key and fits the few KB a tail keeps."""
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06", "bench_C1_driver_flags.json")).read().strip().splitlines()[-1])
    cfg = line["config"]
    flat = bench.summary_scalars(cfg["other_configs"]["runs"], cfg["projected_strong_scaling"]["runs"], line["parity"], cfg["end_to_end"])
    for k in ("c2_msamples", "c3_msamples", "c4_msamples", "c2_frac", "c3_frac", "c4_frac", "c2_parity_ratio", "c3_parity_ratio", "c4_parity_ratio",
              "c1_proj8_msamples", "c3_proj8_msamples", "c4_proj8_msamples", "c1_parity_ratio", "e2e_warm_total_wall_s", "e2e_warm_msamples_whole_command"):
        assert isinstance(flat[k], (int, float)) and flat[k] > 0, k
    assert all(not isinstance(v, (dict, list)) for v in flat.values())
    assert flat["c3_msamples"] == cfg["other_configs"]["runs"]["C3"]["value"] and flat["c1_proj8_msamples"] == cfg["projected_strong_scaling"]["runs"][-1]["value_if_every_gpu_takes_this_long"]
    # the committed line itself: the scalars are in `config`, `summary` repeats them and closes the line
    assert list(line)[-1] == "summary" and len(json.dumps(line["summary"])) < 3000
    for k, v in flat.items():
        assert cfg[k] == v and line["summary"][k] == v, k
    # a failed leg, a leg that did not run
    broken = bench.summary_scalars({"C2": {"error": "x" * 500}, "C3": cfg["other_configs"]["runs"]["C3"]}, [{"error": "boom"}], None, {"error": "no exe"})
    assert broken["c2_error"] == "x" * 100 and "c2_msamples" not in broken and broken["c3_msamples"] == flat["c3_msamples"]
    assert not any(k.startswith(("c1_proj", "c1_parity", "e2e_")) for k in broken)
    assert bench.summary_scalars(None, None, None, None) == {}


def test_the_mean_shift_estimator_sees_a_bias_and_not_the_noise():
    """tools/parity_vs_spp.mean_shift: on synthetic images — independent per-pixel noise around a common mean — the relative shift of
    the mean radiance is within a few standard errors of zero, its standard error is the noise over sqrt(n), and a 1 % bias on top shows as many SE."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parity_vs_spp import mean_shift
    rng = np.random.default_rng(3)
    truth = rng.uniform(0.2, 1.0, (96, 96, 3))
    ref = truth + rng.normal(0, 0.05, truth.shape)
    gpu = truth + rng.normal(0, 0.05, truth.shape)
    mask = np.ones(truth.shape[:2], bool)
    mask[:8] = False
    ms = mean_shift(gpu, ref, mask)
    assert ms["pixels"] == 88 * 96
    assert abs(ms["luminance"]["shift_in_se"]) < 3.5 and 3e-4 < ms["luminance"]["se"] < 2e-3
    biased = mean_shift(gpu * 1.01, ref, mask)
    assert biased["luminance"]["shift_in_se"] > 8 and abs(biased["luminance"]["rel_shift"] - 0.01) < 3 * biased["luminance"]["se"] + 1e-3
    for c in "rgb":
        assert abs(ms[c]["rel_shift"]) < 4 * ms[c]["se"] and biased[c]["rel_shift"] > 5 * biased[c]["se"]
