#!/usr/bin/env python3
"""Developer tool (GPU box): kernel time against launch length (samples per launch) for each forced launch shape —
the fixed cost of a launch and the slope per sample. usage: tools/launch_length.py [scene res shapes]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
shapes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
for shape in shapes:
    os.environ["YHAIR_SHAPE"] = str(shape)
    ctx.init_state(yh.TraceParams.default(resolution=res))
    ctx.trace_samples(64), ctx.trace_samples(64)  # item costs settled
    rows = []
    for spp in (1, 2, 4, 8, 16, 32, 64, 128):
        ms = []
        for _ in range(4):
            ctx.trace_samples(spp)
            ms.append(ctx.last_trace_ms()[0])
        rows.append((spp, min(ms)))
        print(f"{name} {res}^2 shape {shape}: {spp:4d} spp  {min(ms):8.3f} ms  {min(ms) / spp:7.4f} ms per spp", flush=True)
    x, y = np.array([r[0] for r in rows[3:]], float), np.array([r[1] for r in rows[3:]])
    a, b = np.polyfit(x, y, 1)
    print(f"  fit over 8..128 spp: {b:.3f} ms + {a:.4f} ms per spp", flush=True)
