#!/usr/bin/env python3
"""Developer tool (GPU box): A/B of the in-launch pixel hand-over experiment (a library built with -DYH_HANDOVER, tools/build_variants.sh; YHAIR_HANDOVER=1
switches it on from the third launch of a state). usage: YHAIR_LIB=... [YHAIR_HANDOVER=1] tools/handover_ab.py SCENE RES SPP SHAPE LAUNCHES"""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name, res, spp, shape, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
os.environ["YHAIR_SHAPE"] = shape
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=res))
ms = []
for _ in range(n):
    ctx.trace_samples(spp)
    ms.append(ctx.last_trace_ms()[0])
img = ctx.download()
print(f"{name} {res}^2 shape {shape} handover {os.environ.get('YHAIR_HANDOVER', '0')}: ms {np.round(ms, 2)} -> last five {np.mean(ms[-5:]):.2f} ms = "
      f"{res * res * spp / np.mean(ms[-5:]) / 1e3:.0f} Msamples/s, image md5 {hashlib.md5(img.tobytes()).hexdigest()[:8]}", flush=True)
