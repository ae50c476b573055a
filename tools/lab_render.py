#!/usr/bin/env python3
"""Developer tool (GPU box): one config, one forced launch shape, timed launches and the image saved — to set a LAB library variant (YHAIR_LIB=tools/_ab/...) beside
the product when the variant does not render the product's bits (tools/shard_ab.py compares md5s; this one leaves the image for a numeric comparison).
usage: lab_render.py SCENE RES SPP_PER_LAUNCH LAUNCHES SHAPE OUT.npy [RANK WORLD]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh

name, res, spp, launches, shape, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
rank, world = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (0, 1)
kw = {"beta_m": float(os.environ["BETA_M"])} if os.environ.get("BETA_M") else {}
os.environ["YHAIR_NO_DISK_CACHE"] = "1"
os.environ["YHAIR_SHAPE"] = shape
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0, **kw))
ctx.upload_scene(sf.desc)
ctx.set_shard(rank, world)
w, h = ctx.init_state(yh.TraceParams.default(resolution=res))
for _ in range(2): ctx.trace_samples(spp)
ctx.init_state(yh.TraceParams.default(resolution=res))
ms = []
for _ in range(launches):
    ctx.trace_samples(spp)
    ms.append(ctx.last_trace_ms()[0])
img = ctx.download()
np.save(out, img)
tag = os.environ.get("TAG", os.path.basename(os.environ.get("YHAIR_LIB", "product")).replace("libyhair_", "").replace(".so", ""))
px = w * h if world == 1 else None
best = min(ms)
print(f"{tag:10s} {name} {res}^2 shape {shape} rank {rank}/{world}: {best:9.2f} / {float(np.median(ms)):9.2f} ms per {spp} spp"
      + (f" = {px * spp / best / 1e3:8.1f} Msamples/s" if px else ""), flush=True)
sf.close()
