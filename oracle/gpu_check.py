#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle as the checker). Quick end-to-end check of the HIP path against the CPU oracle on one GPU (developer tool;
the judged versions of these checks live in tests/ under -m gpu)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_scenes  # noqa: E402
import oracle_capi as oc  # noqa: E402

yh = oc.yh


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="/tmp/yhair_scenes")
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    o = oc.Oracle()
    ctx = yh.Context(0)
    rng = np.random.default_rng(3)

    # --- unit level: hair BSDF -------------------------------------------------------------
    n = 4096
    mats = np.zeros((n, 12), np.float32)
    mats[:, 3] = rng.uniform(0.1, 0.9, n)   # beta_m
    mats[:, 4] = rng.uniform(0.1, 0.9, n)   # beta_n
    mats[:, 5] = rng.uniform(0, 4, n)       # alpha
    mats[:, 6] = 1.55
    mats[:, 10] = rng.uniform(0, 8, n)      # eumelanin
    mats[: n // 4, 7:10] = rng.uniform(0.05, 0.95, (n // 4, 3))  # colour rows
    v = rng.uniform(0, 1, n).astype(np.float32)
    dirs = lambda k: (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(rng.normal(size=(k, 3))).astype(np.float32)  # noqa
    tng, wo, wi = dirs(n), dirs(n), dirs(n)
    nrm = wo - tng * np.sum(wo * tng, axis=1, keepdims=True)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    rn = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    b_o = o.hair_brdf(mats, v, nrm, tng)
    b_g = ctx.hair_brdf(mats, v, nrm, tng)
    print("brdf max abs diff", np.abs(b_o - b_g).max())
    f_o, f_g = o.hair_eval(b_o, wo, wi), ctx.hair_eval(b_o, wo, wi)
    p_o, p_g = o.hair_pdf(b_o, wo, wi), ctx.hair_pdf(b_o, wo, wi)
    s_o, s_g = o.hair_sample(b_o, wo, rn), ctx.hair_sample(b_o, wo, rn)
    print("eval   rel: median %.2e  p99 %.2e  max %.2e" % (np.median(rel(f_g, f_o)), np.quantile(rel(f_g, f_o), .99), rel(f_g, f_o).max()))
    print("pdf    rel: median %.2e  p99 %.2e  max %.2e" % (np.median(rel(p_g, p_o)), np.quantile(rel(p_g, p_o), .99), rel(p_g, p_o).max()))
    print("sample abs: median %.2e  p99 %.2e  max %.2e" % (np.median(np.abs(s_g - s_o)), np.quantile(np.abs(s_g - s_o), .99), np.abs(s_g - s_o).max()))

    # --- self tests ------------------------------------------------------------------------
    for w in range(4):
        t = time.time()
        ok, worst = ctx.selftest(w)
        print("selftest", w, "OK!" if ok else "TEST FAILED!", "worst %.4g" % worst, "%.2fs" % (time.time() - t))

    # --- scenes ----------------------------------------------------------------------------
    for name, kw, scale in (("sphere-hairblock", {}, 0.02), ("sphere-hairblock", dict(zoom=True), 0.05),
                            ("straight-hair", {}, 0.05), ("curly-hair", {}, 0.05), ("hair-curls", {}, 0.05)):
        path = make_scenes.ensure_scene(name, a.scenes, scale=scale, **kw)
        sf = yh.SceneFile(path)
        osc = o.scene(sf.desc)
        ctx.upload_scene(sf.desc)
        p = yh.TraceParams.default(resolution=96)
        w, h = ctx.init_state(p)
        # closest hits for the camera rays' worth of random rays
        m = 20000
        org = rng.uniform(-1, 1, (m, 3)).astype(np.float32) * 3 + np.array([0, 8 if "hair" in name and name != "sphere-hairblock" else 0.5, 12 if name != "sphere-hairblock" else 4], np.float32)
        tgt = rng.uniform(-1, 1, (m, 3)).astype(np.float32) * np.array([4, 5, 2], np.float32) + np.array([0, 7 if name != "sphere-hairblock" else 0.5, 0], np.float32)
        d = tgt - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([org, d, np.full((m, 1), 1e-4, np.float32), np.full((m, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)
        ho, hg = osc.intersect(rays), ctx.intersect(rays)
        same = all(np.array_equal(x, y) for x, y in zip(ho, hg))
        print(f"{name}{kw}: {m} rays, hits {np.mean(ho[0] >= 0):.3f}, closest-hit bit-identical: {same}")
        if not same:
            bad = np.nonzero((ho[0] != hg[0]) | (ho[1] != hg[1]) | (ho[3] != hg[3]))[0]
            print("   mismatches", len(bad), bad[:5], [x[bad[:3]] for x in ho], [x[bad[:3]] for x in hg])
        for spp in (1, 16):
            ctx.init_state(p)
            t = time.time()
            ctx.trace_samples(spp)
            tg = time.time() - t
            img = ctx.download()
            t = time.time()
            ref = osc.render(p, spp)
            to = time.time() - t
            ident = np.mean(np.all(img == ref, axis=2))
            r = rel(img[..., :3], ref[..., :3]).max(axis=2)
            rmse = np.sqrt(np.mean((img[..., :3] - ref[..., :3]) ** 2)) / max(1e-9, np.mean(ref[..., :3]))
            print(f"   {spp:3d} spp: identical px {ident:.3f}  rel<1e-3 {np.mean(r < 1e-3):.3f}  rel<1e-1 {np.mean(r < 1e-1):.3f}  "
                  f"relRMSE {rmse:.2e}  mean gpu {img[..., :3].mean():.5f} cpu {ref[..., :3].mean():.5f}  gpu {tg*1e3:.1f} ms cpu {to*1e3:.1f} ms")
        osc.close()

    if a.big:
        path = make_scenes.ensure_scene("sphere-hairblock", a.scenes, scale=1.0)
        sf = yh.SceneFile(path)
        t = time.time()
        ctx.upload_scene(sf.desc)
        print("upload C1 scene %.2fs" % (time.time() - t))
        p = yh.TraceParams.default(resolution=720)
        w, h = ctx.init_state(p)
        for spp in (1, 8, 32):
            ctx.trace_samples(spp)
            ms, _ = ctx.last_trace_ms()
            print(f"C1 720x720 {spp} spp: {ms:.1f} ms  -> {w*h*spp/ms/1e3:.1f} Msamples/s")
        wc = ctx.trace_samples_counted(4)
        print("counts", wc.as_dict(), "B/sample %.0f" % wc.bytes_per_sample(4))


if __name__ == "__main__":
    main()
