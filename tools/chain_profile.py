#!/usr/bin/env python3
"""Developer tool (GPU box): the chain of a wave, per forced launch shape — time of a 64-spp launch and the
instrumented launch's wave-level profile (cycles in traversal / shading per wave iteration, trips per iteration,
cycles per trip, which kinds of step ran in how many of the trips).
usage: tools/chain_profile.py SCENE RES SHAPES [WORLD [SPP]]   e.g. sphere-hairblock 180 0,2,4   (WORLD: shard 0 of WORLD only; SPP per launch, default 64)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name, res = sys.argv[1], int(sys.argv[2])
shapes = [int(x) for x in sys.argv[3].split(",")]
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
world = int(sys.argv[4]) if len(sys.argv) > 4 else 1
SPP = int(sys.argv[5]) if len(sys.argv) > 5 else 64
ctx.set_shard(0, world)
if world > 1: name += f" (shard 0 of {world})"
for shape in shapes:
    os.environ["YHAIR_SHAPE"] = str(shape)
    ctx.init_state(yh.TraceParams.default(resolution=res))
    ms = []
    for _ in range(3):
        ctx.trace_samples(SPP)
        ms.append(ctx.last_trace_ms()[0])
    wc = ctx.trace_samples_counted(SPP).as_dict()
    wi = max(1, wc["wave_iters"])
    ws = max(1, wc["wave_steps"])
    print(f"{name} {res}^2 shape {shape}: {SPP} spp {min(ms):.2f} ms (instrumented {ctx.last_trace_ms()[0]:.2f}) | per wave iteration: trace {wc['cyc_trace'] / wi:.0f} cyc, "
          f"shade {wc['cyc_shade'] / wi:.0f} cyc, trips {ws / wi:.2f}, cycles per trip {wc['cyc_trace'] / ws:.0f}, wave iterations {wi} | "
          f"quad trips {wc['lane_steps']}, wave trips {ws}, quad idle trips 1 - quad / (16 x wave) = {1 - wc['lane_steps'] / (16.0 * ws):.3f}, "
          f"live quads per wave iteration {wc['lane_iters'] / wi:.2f}, trips per ray {wc['lane_steps'] / max(1, wc['lane_iters']):.1f} | steps ran in share of trips: "
          + ", ".join(f"{nm} {wc['trips_' + nm] / ws:.2f} ({wc['lanes_' + nm] / max(1, wc['trips_' + nm]):.1f} lanes)" for nm in ("node", "line", "tri", "enter", "scene"))
          + f" | per camera sample: rays {wc['rays'] / max(1, wc['samples']):.2f}, wave trips {ws / max(1, wc['samples']):.3f}, wave iterations {wi / max(1, wc['samples']):.4f}, hair shades {wc['hair_shades'] / max(1, wc['samples']):.3f}, "
          f"surface shades {wc['surf_shades'] / max(1, wc['samples']):.3f}; shading cycles per wave iteration by part: geometry {wc.get('cyc_geom', 0) / wi:.0f}, sample {wc.get('cyc_sample', 0) / wi:.0f}, eval {wc.get('cyc_eval', 0) / wi:.0f}, lights pdf + rest {wc.get('cyc_rest', 0) / wi:.0f}", flush=True)
