#!/usr/bin/env python3
"""Developer tool (GPU box): where k_trace's wave cycles go — traversal vs shading, and inside path_step the hit
geometry, direction sampling (incl. light sampling), BSDF eval + pdf, and the rest (the light pdf with its
instance rays). Counters of the instrumented kernel variant (yh_trace_samples_counted)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import make_scenes, yhair_capi as yh
name = sys.argv[1] if len(sys.argv) > 1 else "sphere-hairblock"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 720
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = yh.Context(0)
sf = yh.SceneFile(make_scenes.ensure_scene(name, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0))
ctx.upload_scene(sf.desc)
ctx.init_state(yh.TraceParams.default(resolution=res))
ctx.trace_samples(spp)
d = ctx.trace_samples_counted(spp).as_dict()
tot = d["cyc_trace"] + d["cyc_shade"]
print(f"{name} {res}^2 x {spp} spp: wave cycles trace {100 * d['cyc_trace'] / tot:.1f} %  shade {100 * d['cyc_shade'] / tot:.1f} %")
ps = d["cyc_geom"] + d["cyc_sample"] + d["cyc_eval"] + d["cyc_rest"]
for k in ("cyc_geom", "cyc_sample", "cyc_eval", "cyc_rest"):
    print(f"  path_step {k[4:]:7s} {100 * d[k] / ps:5.1f} %")
print(f"  per sample: rays {d['rays'] / d['samples']:.2f}, hair shades {d['hair_shades'] / d['samples']:.2f}, surface shades {d['surf_shades'] / d['samples']:.2f}, "
      f"env samples {d['env_samples'] / d['samples']:.2f}; wave steps per ray {d['wave_steps'] * 16 / max(1, d['rays']):.1f} (lane steps {d['lane_steps'] / max(1, d['rays']):.1f})")
for nm in ("node", "line", "tri", "enter", "scene"):
    t, l = d["trips_" + nm], d["lanes_" + nm]
    print(f"  traversal {nm:6s} trips {t:12d}  lanes per trip {l / max(1, t):5.1f}")
