#!/usr/bin/env python3
"""Developer tool (GPU box, a library built with -DYH_LAB_PRIO): does s_setprio move issue slots between the waves
of a SIMD? The same launches with and without raised priority for the n most expensive items; per-item times of the
two runs compared by group (the work of an item is the same in both: same seed). usage: prio_effect.py N [RES]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
os.environ["YHAIR_SHAPE"] = "0"
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene("sphere-hairblock", os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)


def run(prio):
    if prio:
        os.environ["YHAIR_PRIO_ITEMS"] = str(prio)
    else:
        os.environ.pop("YHAIR_PRIO_ITEMS", None)
    ctx.init_state(yh.TraceParams.default(resolution=res))
    out = []
    for _ in range(4):  # the list (and with it the priority switch) is rebuilt after launches 1 and 2
        ctx.trace_samples(64)
        out.append((ctx.last_trace_ms()[0], ctx.item_costs().astype(np.float64) / 100e3))
    return out[-1]


ms_a, a = run(0)
ms_b, b = run(n)
ms_c, c = run(0)
order = np.argsort(a)[::-1]
top, rest = order[:n], order[n:n + 2000]
print(f"launch: plain {ms_a:.2f} ms, priority for the {n} most expensive items {ms_b:.2f} ms, plain again {ms_c:.2f} ms")
for nm, g in (("prioritised items", top), ("the next 2000 items", rest)):
    print(f"  {nm}: mean item time plain {a[g].mean():.3f} ms, with priority {b[g].mean():.3f} ms ({b[g].mean() / a[g].mean():.3f} x), plain again {c[g].mean():.3f} ms ({c[g].mean() / a[g].mean():.3f} x)")
