#!/usr/bin/env python3
"""Developer tool (GPU box, a library built with -DYH_LAB_WHERE): does WHERE a work item ran explain how long it took?
The instrumented kernel stores XCC / SE / SH / CU / SIMD of the wave in the low 13 bits of the item's cost.
usage: YHAIR_LIB=/tmp/yh_sweep/libyhair_where.so tools/where_items_ran.py [SCENE RES SPP]"""
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
for _ in range(4):
    ctx.trace_samples(spp)
    raw = ctx.item_costs()
    runs.append((ctx.last_trace_ms()[0], (raw & ~np.uint32(0x1FFF)).astype(np.float64) / 100e3, raw & 0x1FFF))
for k, (ms, c, w) in enumerate(runs[1:], 1):
    heavy = c > 0.5 * c.max()
    xcc, cu, simd = (w >> 10) & 7, (w >> 2) & 0xFF, w & 3  # cu: SE | SH | CU within the XCC
    place = (w >> 2)  # XCC + CU
    tot = c[heavy].var()
    def explained(key):
        ks = key[heavy]; cs = c[heavy]
        means = {v: cs[ks == v].mean() for v in np.unique(ks)}
        return 1 - np.mean([(x - means[v]) ** 2 for x, v in zip(cs, ks)]) / tot
    print(f"launch {k}: {ms:.2f} ms; {int(heavy.sum())} items above half the top cost, mean {c[heavy].mean():.2f} ms, sd {c[heavy].std():.2f}; variance explained by XCC "
          f"{explained(xcc):.2f}, by CU {explained(place):.2f} ({len(np.unique(place[heavy]))} CUs), by SIMD {explained(w >> 0):.2f}")
    per_x = [c[heavy & (xcc == x)].mean() for x in range(8) if (heavy & (xcc == x)).any()]
    print("   mean heavy-item time per XCC:", " ".join(f"{v:.2f}" for v in per_x))
    pm = np.array([c[heavy & (place == v)].mean() for v in np.unique(place[heavy])])
    print(f"   per-CU mean of heavy items: min {pm.min():.2f}, median {np.median(pm):.2f}, max {pm.max():.2f}; the 12 slowest items ran on CUs {sorted(set(place[np.argsort(-c)[:12]].tolist()))}")
# is it the same CUs every launch?
def per_cu(run):
    ms, c, w = run
    h = c > 0.5 * c.max()
    return {v: c[h & ((w >> 2) == v)].mean() for v in np.unique((w >> 2)[h])}
p1, p2 = per_cu(runs[1]), per_cu(runs[2])
common = sorted(set(p1) & set(p2))
print("correlation of per-CU mean item time between two launches: %.2f" % np.corrcoef([p1[v] for v in common], [p2[v] for v in common])[0, 1])
# the slow XCC: its hardware, or the items it was handed? The same items' times in the PREVIOUS launch (wherever they ran then)
for k in (2, 3):
    ms, c, w = runs[k]
    cp = runs[k - 1][1]
    heavy = c > 0.5 * c.max()
    xcc = (w >> 10) & 7
    print(f"launch {k}: per XCC, mean time of its heavy items now | of the same items in launch {k - 1}: " +
          "  ".join(f"{c[heavy & (xcc == x)].mean():.2f}|{cp[heavy & (xcc == x)].mean():.2f}" for x in range(8) if (heavy & (xcc == x)).any()))
