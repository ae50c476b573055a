#!/usr/bin/env python3
"""Developer tool (GPU box): rays per second of the trace-only kernel (one lane per ray, yh_intersect_batch on a
large batch) at 4 / 5 / 6 / 8 waves per SIMD, on INCOHERENT rays inside a config's hair (origins on the hair,
uniform directions: what the secondary rays of a path look like), next to the quad kernel — the micro-benchmark
behind the question whether a trace stage with a register budget of its own would pay on the dense configs.
usage: tools/trace_only_bench.py SCENE [NRAYS]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "curly-hair"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
d = sf.desc.contents
rng = np.random.default_rng(3)
# origins: random vertices of the line shapes, taken to world space by the frame of an object that shows the shape
pts = []
for oi in range(d.num_objects):
    o = d.objects[oi]
    sh = d.shapes[o.shape]
    if sh.num_lines == 0:
        continue
    pos = np.ctypeslib.as_array(sh.positions, (sh.num_vertices, 3))
    f = np.array(list(o.frame), np.float32).reshape(4, 3)
    pick = pos[rng.integers(0, sh.num_vertices, n // max(1, d.num_objects) + 1)]
    pts.append(pick @ f[:3] + f[3])
org = np.concatenate(pts)[:n].astype(np.float32)
n = len(org)
v = rng.normal(size=(n, 3)).astype(np.float32)
v /= np.linalg.norm(v, axis=1, keepdims=True)
rays = np.concatenate([org, v, np.full((n, 1), 1e-4, np.float32), np.full((n, 1), 3.402823466e+38, np.float32)], axis=1).astype(np.float32)
ref = None
for mode in ("quad", "lane4", "lane5", "lane6", "lane8", "lane5"):
    os.environ["YHAIR_INTERSECT"] = mode
    best = 1e9
    for _ in range(3):
        out = ctx.intersect(rays)
        best = min(best, ctx.last_trace_ms()[0])
    same = "" if ref is None else f"  same hits as quad: {all(np.array_equal(a, b) for a, b in zip(out, ref))}"
    ref = out if ref is None else ref
    print(f"{name}: {n} incoherent rays, {mode:6s}: {best:8.2f} ms = {n / best / 1e3:8.1f} Mrays/s  hit share {np.mean(out[0] >= 0):.3f}{same}", flush=True)
