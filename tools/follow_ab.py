#!/usr/bin/env python3
"""Developer tool (GPU box): how closely the default BSDF arithmetic follows the reference's paths on the reference's own
sphere-hairblock scene (bench.py's parity.path_following_light_hair: rel. RMSE against tests/golden/refscenes.npz over the
seed-to-seed floor), for the library named by YHAIR_LIB — variants of the arithmetic switches of csrc/dev_hair.h
(tools/build_variants.sh, SRC=kernels) — and what the variant costs on C1 with the same kernel (YHAIR_SHAPE, default 0).
usage: YHAIR_LIB=tools/_ab/libyhair_<name>.so tools/follow_ab.py [label]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("YHAIR_SHAPE", "0")
import torch  # noqa
import make_scenes, yhair_capi as yh
scenes = os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes")
ctx = yh.Context(0)
g = np.load(os.path.join(ROOT, "tests", "golden", "refscenes.npz"))
ref8, other = g["sphere-hairblock|8"], g["sphere-hairblock|8_seed777"]
relrmse = lambda a, b: float(np.sqrt(np.mean((a[..., :3] - b[..., :3]) ** 2)) / max(1e-12, np.mean(b[..., :3])))
floor = relrmse(other, ref8)
sf = yh.SceneFile(make_scenes.ensure_scene("ref-sphere-hairblock", scenes, scale=0.05))
ctx.upload_scene(sf.desc)
out = {}
for key, exact in (("fast", False), ("exact", True)):
    ctx.init_state(yh.TraceParams.default(resolution=max(ref8.shape[0], ref8.shape[1]), hair_exact=exact))
    ctx.trace_samples(8)
    img = ctx.download()
    out[key] = relrmse(img, ref8) / floor
    out[key + "_px"] = float(np.mean(np.any(np.abs(img[..., :3] - ref8[..., :3]) > 1e-4 * np.maximum(1e-3, np.abs(ref8[..., :3])), axis=-1)))  # pixels that left the reference's paths
sf.close()
sf = yh.SceneFile(make_scenes.ensure_scene("sphere-hairblock", scenes))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=720))
ms = []
for _ in range(4):
    ctx.trace_samples(64)
    ms.append(ctx.last_trace_ms()[0])
print(f"{sys.argv[1] if len(sys.argv) > 1 else os.environ.get('YHAIR_LIB', 'product'):12s} follow ratio default {out['fast']:.4f} exact {out['exact']:.4f}, pixels off the reference's paths {out['fast_px']:.4f} / {out['exact_px']:.4f} | "
      f"C1 720^2 64 spp shape {os.environ['YHAIR_SHAPE']}: ms {np.round(ms[1:], 2)} -> {720 * 720 * 64 / min(ms[1:]) / 1e3:.0f} Msamples/s", flush=True)
