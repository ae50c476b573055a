#!/usr/bin/env python3
"""Developer tool (GPU box): what ONE GPU of N takes on a BASELINE config — shard 0 of N (tile_id % N == 0) rendered alone, every launch
shape forced in turn, k_stream also with other pool geometries — the table behind profiles/r06/shard_latency_ab.txt.
A pixel's samples are sequential (pt.cpp:1942-1945), so on a shard the launch is per-sample latency x spp once the GPU is under-filled.

usage: shard_ab.py SCENE RES SPP_PER_LAUNCH FULL_SPP WORLDS SHAPES [LAUNCHES]
       e.g. shard_ab.py curly-hair 1280 256 4096 4,8 0,1,3,4,7,6,8
       SHAPES entries: n            launch shape n (YHAIR_SHAPE)
                       3:slots=S    k_stream with S pool slots per wave (YHAIR_ST_SLOTS)
                       3:waves=W    k_stream with at most W waves per CU (YHAIR_ST_WAVES)
       YHAIR_LIB=tools/_ab/libyhair_<variant>.so selects a library variant (tools/build_variants.sh); TAG names it in the output.
Every line: config, shard, shape, ms per launch (min / median of LAUNCHES after two settling launches), seconds for the shard at FULL_SPP,
Msamples/s of the whole image if every GPU takes that long, md5 of the image (equal for every shape and variant: pixels do not depend on the kernel)."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh

name, res, spp, full = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
worlds = [int(x) for x in sys.argv[5].split(",")]
shapes = sys.argv[6].split(",")
launches = int(sys.argv[7]) if len(sys.argv) > 7 else 3
tag = os.environ.get("TAG", os.path.basename(os.environ.get("YHAIR_LIB", "product")).replace("libyhair_", "").replace(".so", ""))
kw = {}
if os.environ.get("BETA_M"): kw["beta_m"] = float(os.environ["BETA_M"])
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0, **kw))
ctx.upload_scene(sf.desc)
os.environ["YHAIR_NO_DISK_CACHE"] = "1"
for world in worlds:
    ctx.set_shard(0, world)
    for sh in shapes:
        for k in ("YHAIR_ST_SLOTS", "YHAIR_ST_WAVES"): os.environ.pop(k, None)
        shape, _, opt = sh.partition(":")
        if opt.startswith("slots="): os.environ["YHAIR_ST_SLOTS"] = opt[6:]
        if opt.startswith("waves="): os.environ["YHAIR_ST_WAVES"] = opt[6:]
        os.environ["YHAIR_SHAPE"] = shape
        try:
            w, h = ctx.init_state(yh.TraceParams.default(resolution=res))
            for _ in range(2): ctx.trace_samples(spp)  # the hand-out order settles (re-planned after launches 1 and 2)
            ctx.init_state(yh.TraceParams.default(resolution=res))
            ms = []
            for _ in range(launches):
                ctx.trace_samples(spp)
                ms.append(ctx.last_trace_ms()[0])
            img = ctx.download()
            md5 = hashlib.md5(img.tobytes()).hexdigest()[:8]
            best, med = min(ms), float(np.median(ms))
            secs = best / spp * full / 1e3
            print(f"{tag:14s} {name} {res}^2 shard 0 of {world} shape {sh:12s}: {best:9.2f} / {med:9.2f} ms per {spp} spp -> {secs:7.3f} s at {full} spp, "
                  f"{w * h * full / secs / 1e6:8.1f} Msamples/s projected at {world} GPUs  md5 {md5}", flush=True)
        except Exception as e:
            print(f"{tag:14s} {name} {res}^2 shard 0 of {world} shape {sh:12s}: FAILED {str(e)[:120]}", flush=True)
sf.close()
