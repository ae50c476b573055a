#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses committed golden images as the checker). Developer tool: how far the
device's paths follow the reference's on each golden scene — share of 1-spp pixels within rel 1e-3 and
relRMSE(gpu, ref) / relRMSE(ref other seed, ref) — for the library named by YHAIR_LIB (A/B of builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python"))
import make_scenes, yhair_capi as yh
from conftest import GOLDEN_SCENES, scene_tag, golden, scene_path
rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
rr = lambda a, b: float(np.sqrt(np.mean((a[..., :3] - b[..., :3]) ** 2)) / max(1e-12, np.mean(b[..., :3])))
ctx = yh.Context(0)
rows = []
g = golden("refscenes.npz")
for which in ("sloth", "bold-man", "straight-hair", "curly-hair", "hair-curls", "sphere-hairblock"):
    rows.append(("ref-" + which, dict(scale=0.05), g[f"{which}|1"], g[f"{which}|8"], g[f"{which}|8_seed777"], 8))
for name, kw in GOLDEN_SCENES:
    s = golden(f"scene_{scene_tag(name, kw)}.npz")
    rows.append((name + str(sorted(kw.items())), (name, kw), s["img_1"], s["img_16"], s["img_16_seed12345"], 16))
for label, spec, r1, rn, ro, n in rows:
    path = scene_path(label, **spec) if isinstance(spec, dict) else scene_path(spec[0], **spec[1])
    sf = yh.SceneFile(path); ctx.upload_scene(sf.desc)
    p = yh.TraceParams.default(resolution=max(r1.shape[0], r1.shape[1]))
    ctx.init_state(p); ctx.trace_samples(1); a = ctx.download()
    ctx.init_state(p); ctx.trace_samples(n); b = ctx.download()
    print(f"{label[:46]:46s} 1 spp within 1e-3: {np.mean(rel(a[..., :3], r1[..., :3]).max(axis=2) < 1e-3):.3f}  identical: {np.mean(np.all(a == r1, axis=2)):.3f}   {n} spp relRMSE ratio: {rr(b, rn) / rr(ro, rn):.3f}")
    sf.close()
