#!/usr/bin/env python3
"""Developer tool (GPU box): the times of a series of equal launches of one state (how the hand-out order's re-planning
settles). usage: tools/launch_series.py [SCENE RES SPP LAUNCHES SHAPE]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
os.environ["YHAIR_SHAPE"] = sys.argv[5] if len(sys.argv) > 5 else "0"
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
for rnd in range(2):
    ctx.init_state(yh.TraceParams.default(resolution=res))
    ms = []
    for _ in range(n):
        ctx.trace_samples(spp)
        ms.append(ctx.last_trace_ms()[0])
    ms = np.array(ms)
    print(f"{name} {res}^2 x {spp} spp x {n} launches: " + " ".join(f"{x:.1f}" for x in ms) + f" | mean of the last {n // 2}: {ms[n // 2:].mean():.2f} ms", flush=True)
