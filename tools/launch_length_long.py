#!/usr/bin/env python3
"""Developer tool (GPU box): ms per sample of ONE launch against its length, up to a whole render in one launch —
what the barrier at the end of every launch costs (each launch waits for its unluckiest pixel)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
os.environ["YHAIR_SHAPE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene("sphere-hairblock", os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=720))
ctx.trace_samples(64), ctx.trace_samples(64)
for spp in (32, 64, 77, 128, 256, 512, 1536):
    ms = []
    for _ in range(3):
        ctx.trace_samples(spp)
        ms.append(ctx.last_trace_ms()[0])
    print(f"C1 720^2 shape {os.environ['YHAIR_SHAPE']}: {spp:5d} spp per launch: {min(ms):9.2f} ms = {min(ms) / spp:.4f} ms per spp (median {np.median(ms) / spp:.4f}) -> {720 * 720 * spp / min(ms) / 1e3:.0f} Msamples/s", flush=True)
