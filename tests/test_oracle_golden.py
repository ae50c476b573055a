"""Pins the CPU oracle (oracle/yh_oracle.cpp) against golden vectors produced by the REAL
reference (oracle/make_golden.py -> tests/golden/*.npz), bit for bit: RNG streams, the hair BSDF
(brdf / eval / sample / pdf), ray-line / ray-triangle / ray-bbox tests, closest hits and whole
rendered images on small synthetic scenes. CPU only."""
import numpy as np
import pytest

from conftest import GOLDEN_SCENES, golden, scene_path, scene_tag


def test_rng_streams(oracle):
    g = golden("rng.npz")
    keys = [k[len("state_"):] for k in g.files if k.startswith("state_")]
    assert len(keys) == 5
    for k in keys:
        seed, seq = (int(x) for x in k.split("_"))
        (st, inc), fl = oracle.rng_stream(seed, seq, 64)
        assert (st, inc) == tuple(int(x) for x in g["state_" + k])
        assert np.array_equal(fl, g["floats_" + k])
    # SURVEY.md 8c sanity values (make_rng(961748941, 1))
    (st, inc), fl = oracle.rng_stream(961748941, 1, 4)
    assert st == 0xefe4fedd9474c793 and inc == 3
    assert np.allclose(fl, [0.893633127, 0.246839881, 0.458433747, 0.477094531], rtol=0, atol=1e-9)
    assert np.array_equal(oracle.pixel_seqs(64 * 64), g["pixel_seqs"])
    assert oracle.pixel_seqs(6).tolist() == [725124800, 678759815, 790335339, 839039096, 650700596, 307940000]


def test_hair_bsdf_bit_exact(oracle):
    g = golden("hair_bsdf.npz")
    brdf = oracle.hair_brdf(g["mats"], g["v"], g["normal"], g["tangent"])
    # -0.0 vs 0.0 in the frame origin is not a difference
    assert np.array_equal(brdf + 0.0, g["brdf"] + 0.0)
    assert np.array_equal(oracle.hair_eval(g["brdf"], g["wo"], g["wi"]), g["f"], equal_nan=True)
    assert np.array_equal(oracle.hair_pdf(g["brdf"], g["wo"], g["wi"]), g["pdf"], equal_nan=True)
    wis = oracle.hair_sample(g["brdf"], g["wo"], g["rn"])
    assert np.array_equal(wis, g["wi_sampled"], equal_nan=True)
    assert np.array_equal(oracle.hair_eval(g["brdf"], g["wo"], wis), g["f_sampled"], equal_nan=True)
    assert np.array_equal(oracle.hair_pdf(g["brdf"], g["wo"], wis), g["pdf_sampled"], equal_nan=True)


def test_hair_bsdf_survey_sanity_row(oracle):
    """The hand-checked row of SURVEY.md 8(c): eumelanin 1.3, defaults, v = 0.7."""
    g = golden("hair_bsdf.npz")
    b = g["brdf"][-1]
    assert np.allclose(b[0:3], [0.544699967, 0.906099975, 1.7809999], rtol=1e-7)
    assert np.isclose(b[5], 0.399999976) and np.isclose(b[17], 0.411516815)
    assert np.allclose(b[6:10], [0.0846111849, 0.0211527962, 0.33844474, 0.33844474], rtol=1e-7)
    assert np.isclose(b[10], 0.117159814)
    assert np.allclose(b[11:14], [0.0348994955, 0.0697564706, 0.139173105], rtol=1e-7)
    assert np.allclose(g["f"][-1], [0.0072593037, 0.00171589118, 0.000117730997], rtol=1e-6)
    assert np.isclose(g["pdf"][-1], 0.0127097657, rtol=1e-6)
    assert np.allclose(g["wi_sampled"][-1], [-0.336120844, -0.789911628, -0.512896121], rtol=1e-6)
    assert np.allclose(g["f_sampled"][-1], [1.45676994, 0.697069347, 0.117038034], rtol=1e-6)
    assert np.isclose(g["pdf_sampled"][-1], 3.73515105, rtol=1e-6)
    # colour row: color (0.8, 0.4, 0.05), beta_n 0.3
    assert np.allclose(g["brdf"][-2][0:3], [0.00143605738, 0.0242141783, 0.258826762], rtol=1e-6)


def test_surface_lobes_bit_exact(oracle, yh):
    """Fresnel terms and the nine surface lobes (yocto_math.h:4215-4755)
    — value * |cos|, pdf and sampled direction — against vectors made by the reference."""
    g = golden("lobes.npz")
    args = (g["params"], g["normal"], g["wo"], g["wi"], g["rn"])
    assert np.array_equal(oracle.fresnel(g["params"], g["normal"], g["wo"]), g["fresnel"], equal_nan=True)
    for kind in range(yh.LOBE_COUNT):
        want = g[f"lobe_{kind}"]
        assert np.array_equal(oracle.surface_lobe(kind, *args), want, equal_nan=True), kind
        # the fixture really exercises the lobe: non-zero values, non-zero samples, finite
        assert np.isfinite(want).all()
        assert (want[:, :3] != 0).any(axis=1).mean() > 0.15, kind
        assert (want[:, 4:] != 0).any(axis=1).mean() > 0.4, kind


def test_curve_conversion_bit_exact(oracle):
    """pbrt curve -> five-vertex strand (yocto_pbrt.h:1751-1797)."""
    g = golden("curves.npz")
    pos, nrm, rad, lines = oracle.curves_to_lines(g["P"], g["width0"], g["width1"], int(g["base_vertex"]))
    assert np.array_equal(pos, g["positions"]) and np.array_equal(nrm, g["normals"], equal_nan=True)
    assert np.array_equal(rad, g["radius"]) and np.array_equal(lines, g["lines"])
    assert lines.min() == 100 and lines.max() == 100 + 5 * len(g["P"]) - 1
    assert np.array_equal(pos[0::5], g["P"][:, 0:3]) and np.array_equal(pos[4::5], g["P"][:, 9:12])


def test_primitive_tests_bit_exact(oracle):
    g = golden("intersect.npz")
    h, uv, d = oracle.intersect_line(g["rays"], g["p0"], g["p1"], g["r0"], g["r1"])
    assert np.array_equal(h, g["line_hit"]) and np.array_equal(uv, g["line_uv"]) and np.array_equal(d, g["line_dist"])
    assert 0.05 < h.mean() < 0.95
    # SURVEY.md 8c sanity ray
    assert h[0] == 1 and np.allclose(uv[0], [0.524203897, 0.946916103], rtol=1e-6) and np.isclose(d[0], 1.00201702)
    h, uv, d = oracle.intersect_triangle(g["rays"], g["p0"], g["p1"], g["p2"])
    assert np.array_equal(h, g["tri_hit"]) and np.array_equal(uv, g["tri_uv"]) and np.array_equal(d, g["tri_dist"])
    assert np.array_equal(oracle.intersect_bbox(g["rays"], g["bbox"]), g["bbox_hit"])


@pytest.mark.parametrize("name,kw", GOLDEN_SCENES, ids=[f"{n}-{'-'.join(map(str, k.values()))}" for n, k in GOLDEN_SCENES])
def test_scene_hits_and_images_bit_exact(oracle, yh, name, kw):
    g = golden(f"scene_{scene_tag(name, kw)}.npz")
    sf = yh.SceneFile(scene_path(name, **kw))
    sc = oracle.scene(sf.desc)
    assert sc.num_lights() == int(g["num_lights"])
    obj, elem, uv, dist = sc.intersect(g["rays"])
    assert np.array_equal(obj, g["object"]) and np.array_equal(elem, g["element"])
    assert np.array_equal(uv, g["uv"]) and np.array_equal(dist, g["distance"])
    res = g["img_1"].shape[0]
    p = yh.TraceParams.default(resolution=res)
    for spp in (1, 16):
        img, rng = sc.render(p, spp, want_rng=True)
        assert np.array_equal(img, g[f"img_{spp}"], equal_nan=True)
        assert np.array_equal(rng, g[f"rng_{spp}"])
    # single-threaded == multi-threaded (the reference is thread-count independent)
    assert np.array_equal(sc.render(p, 16, nthreads=1), g["img_16"])
    assert np.array_equal(sc.render(yh.TraceParams.default(resolution=res, seed=12345), 16), g["img_16_seed12345"])
    sc.close()
    sf.close()


def test_other_shaders_bit_exact(oracle, yh):
    """trace_naive / trace_eyelight / trace_normal (pt.cpp:1514-1672) against the reference's images."""
    g = golden("shaders.npz")
    tags = sorted({k.split("|")[0] for k in g.files})
    by_tag = {scene_tag(n, k): (n, k) for n, k in GOLDEN_SCENES}
    assert len(tags) == 4
    for tag in tags:
        name, kw = by_tag[tag]
        sf = yh.SceneFile(scene_path(name, **kw))
        sc = oracle.scene(sf.desc)
        for shader in ("naive", "eyelight", "normal"):
            p = yh.TraceParams.default(resolution=g[f"{tag}|{shader}|1"].shape[0], shader=shader)
            assert np.array_equal(sc.render(p, 1), g[f"{tag}|{shader}|1"], equal_nan=True)
            img, rng = sc.render(p, 8, want_rng=True)
            assert np.array_equal(img, g[f"{tag}|{shader}|8"], equal_nan=True)
            assert np.array_equal(rng, g[f"{tag}|{shader}|rng8"])
        sc.close(), sf.close()
    # an unknown shader is an error, as in get_trace_shader_func (pt.cpp:1669)
    sf = yh.SceneFile(scene_path(by_tag[tags[0]][0], **by_tag[tags[0]][1]))
    sc = oracle.scene(sf.desc)
    with pytest.raises(Exception):
        sc.render(yh.TraceParams.default(resolution=16, shader=9), 1)
    sc.close(), sf.close()


REF_SCENE_NAMES = ["sloth", "bold-man", "straight-hair", "curly-hair", "hair-curls", "sphere-hairblock"]


@pytest.mark.parametrize("which", REF_SCENE_NAMES)
def test_reference_scene_files_bit_exact(oracle, yh, which):
    """The reference's own scene descriptions (tests/golden/ref_scenes/*.json, verbatim; stand-in
    geometry from tools/make_scenes.py) read by THIS build's loader and rendered by the oracle,
    against images the reference rendered from the same files through its own loader."""
    g = golden("refscenes.npz")
    sf = yh.SceneFile(scene_path("ref-" + which, scale=0.05))
    sc = oracle.scene(sf.desc)
    assert sc.num_lights() == int(g[f"{which}|lights"])
    p = yh.TraceParams.default(resolution=48)
    assert np.array_equal(sc.render(p, 1), g[f"{which}|1"], equal_nan=True)
    img, rng = sc.render(p, 8, want_rng=True)
    assert np.array_equal(img, g[f"{which}|8"], equal_nan=True) and np.array_equal(rng, g[f"{which}|rng8"])
    sc.close(), sf.close()


def test_selftests_pass_on_oracle(oracle):
    """The four Monte-Carlo self-tests of the reference (ext.cpp:555-693) restated; the two
    cheap ones run here, all four run against the GPU in test_gpu_parity.py."""
    for which in (2, 3):
        ok, worst = oracle.selftest(which)
        assert ok, (which, worst)


def test_reference_work_counts_fixture(oracle, yh):
    """the N_* of the algorithmic-bytes formula are counts of the REFERENCE algorithm
    (binary BVH, <= 4 primitives per leaf) made by the instrumented oracle and committed per config
    (tests/golden/workcounts.json, oracle/make_workcounts.py). C0 is re-counted here exactly; the
    formula's bytes/sample of every config is re-derived from its committed counts."""
    import json
    import os
    from conftest import GOLD
    fx = json.load(open(os.path.join(GOLD, "workcounts.json")))
    assert {"C0", "C1", "C2-beta_m0.1", "C2-beta_m0.25", "C2-beta_m0.6", "C3", "C4"} <= set(fx)
    for name, c in fx.items():
        p = c["per_sample"]
        b = (32 * p["nodes"] + 44 * p["seg_tests"] + 52 * p["tri_tests"] + 104 * p["hair_shades"] +
             (48 * p["env_lookups"] if c["env_textured"] else 0) + 88 * p["env_samples"] + 32.0 / c["spp_per_launch_assumed"])  # env texels: textured environments only
        assert abs(b - c["algorithmic_bytes_per_sample"]) < 0.6, name
    c0 = fx["C0"]
    sf = yh.SceneFile(scene_path("sphere-hairblock"))
    osc = oracle.scene(sf.desc)
    _, wc = osc.render(yh.TraceParams.default(resolution=c0["resolution"]), c0["spp_counted"], want_counts=True)
    got = {k: round(v / wc.samples, 4) for k, v in wc.as_dict().items() if k in c0["per_sample"]}
    assert got == c0["per_sample"]
    osc.close(), sf.close()
