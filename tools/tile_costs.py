#!/usr/bin/env python3
"""Developer tool: per-tile cost distribution of one launch (load-balance analysis)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_scenes, yhair_capi as yh
scene = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(scene, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
w, h = ctx.init_state(yh.TraceParams.default(resolution=720))
for n in (1, 4, 16, spp, spp):
    ctx.trace_samples(n)
    ms, _ = ctx.last_trace_ms()
    c = ctx.tile_costs().astype(np.float64) / 100e3  # ms
    s = np.sort(c.ravel())[::-1]
    print(f"{n:3d} spp: kernel {ms:8.2f} ms | tile cost ms: max {s[0]:.2f} p99 {s[len(s)//100]:.2f} median {np.median(s):.3f} sum {s.sum():.1f} "
          f"| sum/2048 waves {s.sum()/2048:.2f} | top5 {np.round(s[:5],2)} | tiles>1ms {np.sum(s>1)}")
t0 = time.time(); wc = ctx.trace_samples_counted(64).as_dict(); print("instrumented 64 spp launch: %.1f ms" % ctx.last_trace_ms()[0])
print("profile (64 spp, instrumented):", wc)
wi = max(1, wc["wave_iters"])
print(f"  per wave-iteration: trace {wc['cyc_trace']/wi:.0f} cyc, shade {wc['cyc_shade']/wi:.0f} cyc, traversal trips {wc['wave_steps']/wi:.1f} (lane avg {wc['lane_steps']/max(1,wc['lane_iters']):.1f}), live lanes {wc['lane_iters']/wi:.1f}; cycles per trip {wc['cyc_trace']/max(1,wc['wave_steps']):.0f}")
print("  traversal divergence (code executed in X of the wave trips, with Y of 64 lanes active):",
      {nm: (round(wc["trips_" + nm] / max(1, wc["wave_steps"]), 3), round(wc["lanes_" + nm] / max(1, wc["trips_" + nm]), 1)) for nm in ("node", "line", "tri", "enter", "scene")})
print("  shade split (lane-0 cycles):", {k: round(wc[k] / max(1, wc["cyc_geom"] + wc["cyc_sample"] + wc["cyc_eval"] + wc["cyc_rest"]), 3) for k in ("cyc_geom", "cyc_sample", "cyc_eval", "cyc_rest")})
top = np.argsort(c.ravel())[::-1][:8]
print("heaviest tiles (ty,tx):", [(int(t // c.shape[1]), int(t % c.shape[1])) for t in top])
img = ctx.download()
hitfrac = img[..., 3].reshape(h // 8, 8, w // 8, 8).mean(axis=(1, 3))
print("corr(cost, hit fraction) = %.3f" % np.corrcoef(c.ravel(), hitfrac.ravel())[0, 1])
rows = c.sum(axis=1)
print("cost by tile row:", np.round(rows, 1))
