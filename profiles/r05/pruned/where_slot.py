#!/usr/bin/env python3
"""Developer tool (GPU box, a library built with -DYH_LAB_WHERE -DYH_LAB_WHERE_SLOT): item time against the wave's age —
which half of the grid its workgroup is in, which wave of the workgroup it is, which hardware wave slot it got.
usage: YHAIR_LIB=... tools/where_slot.py [SCENE RES SPP]"""
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
for k in range(4):
    ctx.trace_samples(spp)
    raw = ctx.item_costs()
    c, w = (raw & ~np.uint32(0x1FFF)).astype(np.float64) / 100e3, raw & 0x1FFF
    if k == 0:
        continue
    h = c > 0.5 * c.max()
    half, wib, slot, simd = (w >> 9) & 1, (w >> 6) & 7, (w >> 2) & 15, w & 3
    print(f"launch {k}: {ctx.last_trace_ms()[0]:.2f} ms, {int(h.sum())} heavy items, mean {c[h].mean():.2f}")
    print("   by half of the grid (0 = workgroups dispatched first):", " ".join(f"{c[h & (half == v)].mean():.2f} ({int((h & (half == v)).sum())})" for v in (0, 1)))
    print("   by wave of the workgroup:", " ".join(f"{c[h & (wib == v)].mean():.2f}" for v in range(8)))
    print("   by hardware wave slot:", " ".join(f"{v}:{c[h & (slot == v)].mean():.2f}({int((h & (slot == v)).sum())})" for v in range(16) if (h & (slot == v)).any()))
    top = np.argsort(-c)[:16]
    print("   the 16 slowest items: (half, wave of workgroup, slot):", " ".join(f"({half[i]},{wib[i]},{slot[i]})" for i in top))
