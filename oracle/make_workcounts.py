#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Work counts of the REFERENCE ALGORITHM per BASELINE config:
the instrumented CPU oracle (bit-identical to the reference) renders each config's full-size
synthetic scene at its full resolution for a few samples per pixel and the per-sample averages —
the N_* of the algorithmic-bytes formula — are committed as tests/golden/workcounts.json.

    python oracle/make_workcounts.py            # all configs (C3 / C4 take a minute on 8 cores)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_scenes  # noqa: E402
import oracle_capi as oc  # noqa: E402

yh = oc.yh
CONFIGS = [  # name, scene, overrides, resolution, spp of the config, spp rendered here, spp per launch of the GPU path
    ("C0", "sphere-hairblock", {}, 256, 64, 4, 64),
    ("C1", "sphere-hairblock", {}, 720, 1536, 2, 64),
    ("C2-beta_m0.1", "straight-hair", {"beta_m": 0.1}, 720, 1536, 1, 64),
    ("C2-beta_m0.25", "straight-hair", {"beta_m": 0.25}, 720, 1536, 1, 64),
    ("C2-beta_m0.6", "straight-hair", {"beta_m": 0.6}, 720, 1536, 1, 64),
    ("C3", "curly-hair", {}, 1280, 4096, 1, 32),
    ("C4", "hair-curls", {}, 1280, 4096, 1, 32),
]


def main():
    only = sys.argv[1:]
    out_path = os.path.join(ROOT, "tests", "golden", "workcounts.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    oracle = oc.Oracle()
    for name, scene, kw, res, spp, spp_here, spp_launch in CONFIGS:
        if only and name not in only:
            continue
        path = make_scenes.ensure_scene(scene, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0, **kw)
        sf = yh.SceneFile(path)
        d = sf.desc.contents
        segs = sum(d.shapes[d.objects[i].shape].num_lines for i in range(d.num_objects))
        osc = oracle.scene(sf.desc)
        t = time.time()
        img, wc = osc.render(yh.TraceParams.default(resolution=res), spp_here, want_counts=True)
        dt = time.time() - t
        s = wc.samples
        per = {k: round(v / s, 4) for k, v in wc.as_dict().items()
               if k in ("rays", "nodes", "seg_tests", "tri_tests", "hair_shades", "surf_shades", "env_lookups", "env_samples")}
        env_tex = any(d.environments[i].tex_width > 0 for i in range(d.num_environments))  # env texels count for a textured environment only
        out[name] = {"scene": scene, "overrides": kw, "env_textured": env_tex, "hair_segments_instanced": int(segs), "resolution": res,
                     "image": [int(img.shape[1]), int(img.shape[0])], "spp_of_config": spp, "spp_counted": spp_here,
                     "per_sample": per, "algorithmic_bytes_per_sample": round(wc.bytes_per_sample(spp_launch, env_tex), 1),
                     "spp_per_launch_assumed": spp_launch,
                     "oracle_msamples_per_s_here": round(s / dt / 1e6, 3), "host_threads_here": os.cpu_count()}
        print(name, out[name]["per_sample"], out[name]["algorithmic_bytes_per_sample"], "B/sample,", f"{dt:.1f} s")
        osc.close(), sf.close()
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
