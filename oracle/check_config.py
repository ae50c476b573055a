#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle as the checker). Runs one BASELINE config at full scene
size on the GPU: parity against the oracle at a reduced resolution (same full-size geometry) and
throughput at the config's resolution."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, oracle_capi as oc
yh = oc.yh
ap = argparse.ArgumentParser()
ap.add_argument("scene"); ap.add_argument("--res", type=int, default=720); ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--steps", type=int, default=4); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--beta-m", type=float, default=None); ap.add_argument("--check-res", type=int, default=96)
a = ap.parse_args()
kw = {} if a.beta_m is None else {"beta_m": a.beta_m}
path = make_scenes.ensure_scene(a.scene, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=a.scale, **kw)
ctx = yh.Context(0)
sf = yh.SceneFile(path)
d = sf.desc.contents
segs = sum(d.shapes[i].num_lines for i in range(d.num_shapes))
t = time.time(); ctx.upload_scene(sf.desc); up = time.time() - t
print(f"{a.scene}: {segs} segments, {d.num_objects} objects, upload+BVH {up:.2f}s")
# parity at reduced resolution, full geometry
osc = oc.Oracle().scene(sf.desc)
p = yh.TraceParams.default(resolution=a.check_res)
ctx.init_state(p); ctx.trace_samples(1); g1 = ctx.download(); r1 = osc.render(p, 1)
rel = np.abs(g1[..., :3] - r1[..., :3]) / np.maximum(np.abs(r1[..., :3]), 1e-6)
print(f"  1 spp {a.check_res}^2: identical px {np.mean(np.all(g1 == r1, axis=2)):.3f}, rel<1e-3 {np.mean(rel.max(axis=2) < 1e-3):.3f}, hit frac {r1[..., 3].mean():.3f}")
ctx.init_state(p); ctx.trace_samples(16); g = ctx.download(); r = osc.render(p, 16)
r2 = osc.render(yh.TraceParams.default(resolution=a.check_res, seed=12345), 16)
rr = lambda x, y: np.sqrt(np.mean((x[..., :3] - y[..., :3]) ** 2)) / np.mean(y[..., :3])
print(f"  16 spp: relRMSE gpu-vs-oracle {rr(g, r):.4f}, seed floor {rr(r2, r):.4f}, mean gpu {g[..., :3].mean():.5f} oracle {r[..., :3].mean():.5f}")
_, wc = osc.render(p, 2, want_counts=True)
print(f"  reference work/sample: {({k: round(v / wc.samples, 2) for k, v in wc.as_dict().items() if k in ('rays','nodes','seg_tests','tri_tests','hair_shades','env_samples')})}  B/sample {wc.bytes_per_sample(a.spp):.0f}")
# throughput
p = yh.TraceParams.default(resolution=a.res)
w, h = ctx.init_state(p)
for s in range(a.steps):
    ctx.trace_samples(a.spp)
    ms, _ = ctx.last_trace_ms()
    print(f"  step {s}: {a.spp} spp {w}x{h}: {ms:.1f} ms -> {w * h * a.spp / ms / 1e3:.1f} Msamples/s")
img = ctx.download()
print(f"  image finite {np.isfinite(img).all()}, mean {img[..., :3].mean(axis=(0, 1))}, hit frac {(img[..., 3] > 0).mean():.3f}")
