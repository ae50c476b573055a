#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle as the checker). Developer tool: error distribution of the device hair BSDF against the reference's golden vectors
(tests/golden/hair_bsdf.npz): max / 99.9th percentile relative error of f and pdf, absolute error
of sampled directions, and the share of samples whose lobe choice differs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_capi as oc
yh = oc.yh
g = np.load(os.path.join(ROOT, "tests", "golden", "hair_bsdf.npz"))
ctx = yh.Context(0)
rel = lambda a, b, fl: np.abs(a - b) / np.maximum(np.abs(b), fl)
def report(name, e):
    e = e[np.isfinite(e)]
    print(f"{name:28s} max {e.max():.3e}  p99.9 {np.quantile(e, 0.999):.3e}  p99 {np.quantile(e, 0.99):.3e}  median {np.median(e):.3e}  >1e-4: {np.mean(e > 1e-4):.5f}")
b = ctx.hair_brdf(g["mats"], g["v"], g["normal"], g["tangent"])
report("hair_brdf fields (rel)", rel(b, g["brdf"], 1e-3).ravel())
ok = np.isfinite(g["f"]).all(axis=1) & np.isfinite(g["pdf"])
f, pdf = ctx.hair_eval(g["brdf"], g["wo"], g["wi"]), ctx.hair_pdf(g["brdf"], g["wo"], g["wi"])
report("f (rel, floor 1e-7)", rel(f[ok], g["f"][ok], 1e-7).max(axis=1))
report("pdf (rel, floor 1e-7)", rel(pdf[ok], g["pdf"][ok], 1e-7))
wi = ctx.hair_sample(g["brdf"], g["wo"], g["rn"])
okd = np.isfinite(g["wi_sampled"]).all(axis=1)
d = np.abs(wi[okd] - g["wi_sampled"][okd]).max(axis=1)
report("sampled direction (abs)", d)
print("   directions off by > 1e-2 (another lobe chosen):", int(np.sum(d > 1e-2)), "of", len(d))
worst = np.argsort(rel(pdf[ok], g["pdf"][ok], 1e-7))[-5:]
print("   worst pdf rows: beta_m", g["mats"][ok][worst][:, 3], "values", g["pdf"][ok][worst], "got", pdf[ok][worst])

# fresh random inputs against the oracle (as tests/test_gpu_parity.py does, 4x the rows)
rng = np.random.default_rng(11)
n = 200000
mats = np.zeros((n, 12), np.float32)
mats[:, 3:5] = rng.uniform(0.05, 0.95, (n, 2)); mats[:, 5] = rng.uniform(0, 4, n); mats[:, 6] = 1.55; mats[:, 10] = rng.uniform(0, 8, n)
v = rng.uniform(0, 1, n).astype(np.float32)
dd = lambda: (lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))(rng.normal(size=(n, 3)))
tng, wo, wi, nrm = dd(), dd(), dd(), dd()
o = oc.Oracle()
bb = o.hair_brdf(mats, v, nrm, tng)
fo, fg = o.hair_eval(bb, wo, wi), ctx.hair_eval(bb, wo, wi)
po, pg = o.hair_pdf(bb, wo, wi), ctx.hair_pdf(bb, wo, wi)
report("fresh f (rel, floor 1e-7)", rel(fg, fo, 1e-7).max(axis=1))
report("fresh pdf (rel, floor 1e-7)", rel(pg, po, 1e-7))
e = rel(pg, po, 1e-7); w = np.argsort(np.nan_to_num(e))[-5:]
print("   worst fresh pdf rows: beta_m", mats[w, 3], "beta_n", mats[w, 4], "oracle", po[w], "gpu", pg[w])
