#!/usr/bin/env python3
"""Developer tool (GPU box): the most expensive work items of consecutive launches — are they the same items, and how
far above the rest? usage: tools/top_items.py [SCENE RES SPP SHAPE]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
os.environ["YHAIR_SHAPE"] = sys.argv[4] if len(sys.argv) > 4 else "0"
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=res))
tx = (res + 7) // 8
for k in range(5):
    ctx.trace_samples(spp)
    ms = ctx.last_trace_ms()[0]
    c = ctx.item_costs().astype(np.float64) / 100e3
    top = np.argsort(c)[::-1][:12]
    print(f"launch {k}: {ms:.2f} ms; top items (tile x, tile y, quadrant: ms): " + "  ".join(f"({(t >> 2) % tx},{(t >> 2) // tx},{t & 3}: {c[t]:.2f})" for t in top), flush=True)
