#!/usr/bin/env python3
"""Developer tool (GPU box): how repeatable is a work item's cost from one launch to the next (same kernel, same image,
the next 64 samples)? usage: tools/item_cost_repeat.py [SCENE RES SPP]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
os.environ["YHAIR_SHAPE"] = "0"
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=res))
runs = []
for _ in range(6):
    ctx.trace_samples(spp)
    runs.append((ctx.last_trace_ms()[0], ctx.item_costs().astype(np.float64) / 100e3))
for k in range(2, 6):
    (ma, a), (mb, b) = runs[k - 1], runs[k]
    h = a > 0.5 * a.max()
    r = b[h] / a[h]
    top = np.argsort(-b)[:8]
    print(f"launch {k}: {mb:.2f} ms (previous {ma:.2f}); items above half the previous top: {int(h.sum())}, correlation of their costs with the previous launch {np.corrcoef(a[h], b[h])[0, 1]:.2f}, "
          f"ratio now / before: median {np.median(r):.2f}, 1 % {np.quantile(r, 0.01):.2f}, 99 % {np.quantile(r, 0.99):.2f}, max {r.max():.2f}")
    print("   the 8 slowest items now (ms now | ms before): " + "  ".join(f"{b[i]:.1f}|{a[i]:.1f}" for i in top))
mean = np.mean([c for _, c in runs[1:]], axis=0)
h = mean > 0.5 * mean.max()
noise = np.std([c[h] for _, c in runs[1:]], axis=0)
print(f"over launches 1-5: persistent spread of the item means sd {mean[h].std():.2f} ms (mean {mean[h].mean():.2f}); launch-to-launch noise of one item sd {np.median(noise):.2f} ms (median over items)")
