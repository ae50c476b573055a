#!/usr/bin/env python3
"""Developer tool (GPU box): distribution of the work items' costs of one launch — how far the most expensive items
(which bound a chain-bound launch) are from the typical expensive item.
usage: tools/item_cost_distribution.py [SCENE RES SPP SHAPE]"""
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
for _ in range(3):
    ctx.trace_samples(spp)
ms = ctx.last_trace_ms()[0]
c = ctx.item_costs().astype(np.float64) / 100e3  # ms per work item, 100 MHz ticks
s = np.sort(c)[::-1]
heavy = s[s > 0.2 * s[0]]
print(f"{name} {res}^2 x {spp} spp shape {os.environ['YHAIR_SHAPE']}: launch {ms:.2f} ms; {len(s)} items, {len(heavy)} within 5x of the most expensive")
print("  item cost (ms): max %.2f | top 1 %% %.2f | 5 %% %.2f | 10 %% %.2f | 15 %% %.2f | 25 %% %.2f | 50 %% %.2f | 75 %% %.2f of the expensive items | mean %.2f" % (
    heavy[0], *[heavy[int(len(heavy) * q)] for q in (0.01, 0.05, 0.10, 0.15, 0.25, 0.5, 0.75)], heavy.mean()))
print("  wave-time of all items / 4096 slots: %.2f ms (launch %.2f ms)" % (c.sum() / 4096, ms))
