#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle for reference work counts). Developer tool: camera-ray traversal micro-benchmark (k_intersect) on the C1 scene. Run under
rocprofv3 --kernel-trace to get per-dispatch durations; prints the reference work counts of each
ray set so that cycles per node / per wave-trip can be derived."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, oracle_capi as oc
yh = oc.yh
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene("sphere-hairblock", os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
osc = oc.Oracle().scene(sf.desc)
cam = sf.desc.contents.camera
F = np.array(cam.frame[:], np.float32).reshape(4, 3)
def camera_rays(px, py, W=720, H=720):
    u = (px + 0.5) / W; v = (py + 0.5) / H
    q = np.stack([cam.film[0] * (0.5 - u), cam.film[1] * (v - 0.5), np.full_like(u, cam.lens)], 1).astype(np.float32)
    dc = -q / np.linalg.norm(q, axis=1, keepdims=True)
    d = dc @ F[:3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(F[3], (len(px), 1))
    return np.concatenate([o, d, np.full((len(px), 1), 1e-4), np.full((len(px), 1), 3.4e38)], 1).astype(np.float32)
def tile_rays(tx, ty):
    ii, jj = np.meshgrid(np.arange(8), np.arange(8))
    return camera_rays((tx * 8 + ii).ravel().astype(np.float32), (ty * 8 + jj).ravel().astype(np.float32))
rs = np.random.default_rng(1)
def inside_rays(n):  # bounce-like rays: origins inside the hair block (world [0,1]x[0,1]x[-1,0]), random directions
    o = rs.uniform(0.02, 0.98, (n, 3)).astype(np.float32) * [1, 1, 1] + [0, 0, -1]
    d = rs.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d, np.full((n, 1), 1e-4), np.full((n, 1), 3.4e38)], 1).astype(np.float32)
sets = {
    "inside hair block 64 rays": inside_rays(64),
    "inside hair block 65536 rays": inside_rays(65536),
    "inside hair block 1M rays": inside_rays(1 << 20),
    "heavy tile (38,51) 64 rays": tile_rays(38, 51),
    "mid hair tile (45,44) 64 rays": tile_rays(45, 44),
    "hair region 27x27 tiles": np.concatenate([tile_rays(tx, ty) for ty in range(30, 57) for tx in range(33, 60)]),
    "full frame 518400 rays": np.concatenate([tile_rays(tx, ty) for ty in range(90) for tx in range(90)]),
}
for name, rays in sets.items():
    nodes, prims = osc.intersect_counted(rays)
    w = len(rays) // 64
    nm = nodes.reshape(w, 64)
    t = time.time(); ctx.intersect(rays); dt = time.time() - t
    print(f"{name}: rays {len(rays)} nodes/ray {nodes.mean():.1f} prims/ray {prims.mean():.1f} | per wave: sum(max nodes) {nm.max(1).sum()} "
          f"simd eff {nodes.sum() / (64.0 * nm.max(1).sum()):.2f} | host wall {dt*1e3:.2f} ms")
