#!/usr/bin/env python3
"""Developer tool (GPU box): the sample-loop kernels against each other (YHAIR_SHAPE forces one: 0, 1 k_trace over
4-wide nodes, 2 k_trace over 8-wide nodes, 3 k_stream, 4 k_trace with octets) — bitwise image / RNG comparison on
small scenes, then throughput of each on a BASELINE config.
usage: shape_check.py check            (WF_SHAPE=n against shape 0 on seven small scenes)
       shape_check.py SCENE RES SPP SHAPES [WORLD]   e.g. sphere-hairblock 720 64 0,2,4   (WORLD: shard 0 of WORLD only)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
ctx = yh.Context(0)
SCENES = os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes")


def render(path_or_sf, res, spps, shape):
    os.environ["YHAIR_SHAPE"] = str(shape)
    ctx.init_state(yh.TraceParams.default(resolution=res))
    ms = []
    for n in spps:
        ctx.trace_samples(n)
        ms.append(ctx.last_trace_ms()[0])
    return ctx.download(), ctx.download_rng(), ms


if len(sys.argv) > 1 and sys.argv[1] == "check":
    for name, kw, res in [("sphere-hairblock", dict(scale=0.05, zoom=True), 88), ("straight-hair", dict(scale=0.05), 64),
                          ("hair-curls", dict(scale=0.05), 50), ("lobes", dict(scale=0.05), 72), ("volumes", dict(scale=0.05), 64),
                          ("crowd", dict(scale=0.05), 64), ("textured", dict(scale=0.05), 64)]:
        sf = yh.SceneFile(make_scenes.ensure_scene(name, SCENES, **kw))
        ctx.upload_scene(sf.desc)
        a, ra, _ = render(sf, res, (1, 3, 4), 0)
        b, rb, _ = render(sf, res, (1, 3, 4), int(os.environ.get("WF_SHAPE", "2")))
        print(f"{name:18s} res {res}: images equal {np.array_equal(a, b)}  rng equal {np.array_equal(ra, rb)}  "
              f"max |d| {np.abs(a - b).max():.3g}  differing px {int(np.any(a != b, axis=2).sum())}", flush=True)
        sf.close()
else:
    name = sys.argv[1] if len(sys.argv) > 1 else "straight-hair"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    shapes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2").split(",")]
    sf = yh.SceneFile(make_scenes.ensure_scene(name, SCENES, scale=1.0))
    ctx.upload_scene(sf.desc)
    world = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    ctx.set_shard(0, world)
    if world > 1: name += f" (shard 0 of {world})"
    ref = None
    for rnd in range(2):
        for shape in shapes:
            img, _, ms = render(sf, res, (spp, spp, spp), shape)
            same = f" image md5 {__import__('hashlib').md5(img.tobytes()).hexdigest()[:8]}" if ref is None else f" same image as first: {np.array_equal(img, ref)}"
            ref = img if ref is None else ref
            print(f"{name} {res}^2 x {spp} spp shape {shape}: ms {np.round(ms, 2)} -> "
                  f"{res * res * spp / ms[-1] / 1e3:.1f} Msamples/s{same}", flush=True)
