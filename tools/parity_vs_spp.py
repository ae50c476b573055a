#!/usr/bin/env python3
"""Developer tool (GPU box): a dense config against the reference AT THE CONFIG'S OWN SAMPLE COUNT, with a bias estimator.

The path-following ratio relRMSE(gpu, ref) / relRMSE(ref seed A, ref seed B) grows with spp (0.14 at 8 spp, 0.25 at 64 on C3): what per-pixel stream
DECORRELATION predicts — a device path that leaves the reference's (a last-place difference in a libm result flips a lobe choice) shifts every
later draw of that pixel's one PCG32 stream, so the share of a pixel's samples that still follow the reference's falls with the sample count and
the ratio tends to 1, the score of an exact but independent renderer. A BIAS of the fast BSDF arithmetic would look the same in that ratio. What
separates them: decorrelated estimates have the same MEAN, so the mean radiance over the pixels that see the model differs by sampling noise
only — a difference whose standard error falls like 1 / sqrt(spp) — while a bias is a mean shift that stays when the samples grow.

usage: parity_vs_spp.py CONFIG RES [SPPS]     CONFIG: C2 | C3 | C4; RES: a reduced image side (the config's camera and geometry); SPPS: 8,64,512,full
For each spp: the reference (oracle/_ref/libyh_ref.so on this host's threads, else the bit-identical oracle) at seeds A and B, the device at both seeds
with the default and with the exact BSDF arithmetic; per spp: ratio to floor, share of pixels within 4 sigma, and the relative difference of the
mean radiance over the hit pixels with its standard error (per channel and luminance; both seeds pooled). Prints one JSON object."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import make_scenes, oracle_capi as oc, yhair_capi as yh

CONFIGS = {"C2": ("straight-hair", {"beta_m": 0.25}, 1536), "C3": ("curly-hair", {}, 4096), "C4": ("hair-curls", {}, 4096)}
K_SIGMA = 4.0


def mean_shift(gpu, ref, mask):
    """Relative difference of the mean radiance over the masked pixels, with its standard error: the per-pixel differences are independent draws
    (every pixel has its own stream), so SE = std(diff) / sqrt(n). gpu, ref: (H, W, 3) float64."""
    d = (gpu - ref)[mask]  # (n, 3)
    n = d.shape[0]
    base = ref[mask].mean(axis=0)
    lum = np.array([0.2126, 0.7152, 0.0722])
    dl, bl = d @ lum, float(ref[mask].mean(axis=0) @ lum)
    out = {"pixels": int(n)}
    for k, c in enumerate("rgb"):
        out[c] = {"rel_shift": float(d[:, k].mean() / base[k]), "se": float(d[:, k].std(ddof=1) / np.sqrt(n) / base[k])}
    out["luminance"] = {"rel_shift": float(dl.mean() / bl), "se": float(dl.std(ddof=1) / np.sqrt(n) / bl)}
    out["luminance"]["shift_in_se"] = out["luminance"]["rel_shift"] / max(out["luminance"]["se"], 1e-30)
    return out


def compare(ga, gb, ca, cb):
    ga, gb, ca, cb = (x[..., :3].astype(np.float64) for x in (ga, gb, ca, cb))
    mean = float(ca.mean())
    rel = float(np.sqrt(np.mean((ga - ca) ** 2)) / mean)
    floor = float(np.sqrt(np.mean((cb - ca) ** 2)) / mean)
    var = np.stack([ga, gb, ca, cb]).var(axis=0, ddof=1)
    ok = np.abs(ga - ca) <= K_SIGMA * np.sqrt(2.0 * var) + 1e-3 * np.abs(ca) + 1e-6
    return {"rel_rmse_gpu_vs_ref": rel, "rel_rmse_ref_seed_floor": floor, "ratio_to_floor": rel / floor if floor > 0 else None,
            "share_within_4_sigma": float(ok.all(axis=2).mean())}


def main():
    cfg, res = sys.argv[1], int(sys.argv[2])
    scene, kw, full = CONFIGS[cfg]
    spps = [full if s == "full" else int(s) for s in (sys.argv[3] if len(sys.argv) > 3 else "8,64,512,full").split(",")]
    path = make_scenes.ensure_scene(scene, os.environ.get("YHAIR_SCENES", "/tmp/yhair_scenes"), scale=1.0, **kw)
    threads = os.cpu_count() or 1
    sf = yh.SceneFile(path)
    if oc.have_ref():
        rsc, kind = oc.Ref().scene(path), "reference (oracle/_ref/libyh_ref.so)"
        render = lambda p, n: rsc.render(p, n)
    else:
        osc, kind = oc.Oracle().scene(sf.desc), "oracle (bit-identical port)"
        render = lambda p, n: osc.render(p, n, nthreads=threads)
    ctx = yh.Context(0)
    ctx.upload_scene(sf.desc)
    seed_b = 12345
    out = {"config": cfg, "scene": scene, "overrides": kw, "resolution": res, "full_spp": full, "cpu": kind, "cpu_threads": threads, "seeds": [961748941, seed_b],
           "what": "per spp: the device image against the reference's at equal seed (ratio_to_floor: relRMSE over the reference's own seed-to-seed relRMSE; share of pixels "
                   "within 4 sigma), and mean_shift: (mean of gpu - mean of ref) / mean of ref over the pixels that see the model, both seeds pooled, with its standard "
                   "error — a bias is a shift that does not shrink with spp, decorrelation is not a shift at all", "runs": []}
    for spp in spps:
        t0 = time.time()
        pa, pb = yh.TraceParams.default(resolution=res), yh.TraceParams.default(resolution=res, seed=seed_b)
        ca = render(pa, spp)
        print(f"[parity_vs_spp] {cfg} {res}^2 {spp} spp: reference seed A {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        cb = render(pb, spp)
        cpu_s = time.time() - t0
        print(f"[parity_vs_spp] {cfg} {res}^2 {spp} spp: reference seed B, {cpu_s:.1f} s both", file=sys.stderr, flush=True)
        run = {"spp": spp, "cpu_seconds_two_seeds": round(cpu_s, 1), "cpu_msamples_per_s": round(2 * ca.shape[0] * ca.shape[1] * spp / cpu_s / 1e6, 3)}
        mask = (ca[..., 3] > 0) & (cb[..., 3] > 0)
        for key, exact in (("fast_bsdf", False), ("exact_bsdf", True)):
            imgs = []
            for seed in (None, seed_b):
                p = yh.TraceParams.default(resolution=res, hair_exact=exact) if seed is None else yh.TraceParams.default(resolution=res, seed=seed, hair_exact=exact)
                ctx.init_state(p)
                done = 0
                while done < spp:  # (launches of at most 512 spp)
                    n = min(512, spp - done)
                    ctx.trace_samples(n)
                    done += n
                imgs.append(ctx.download())
            r = compare(imgs[0], imgs[1], ca, cb)
            r["alpha_identical"] = bool(np.array_equal(imgs[0][..., 3] > 0, ca[..., 3] > 0))
            g = (imgs[0][..., :3].astype(np.float64) + imgs[1][..., :3].astype(np.float64)) / 2
            c = (ca[..., :3].astype(np.float64) + cb[..., :3].astype(np.float64)) / 2
            r["mean_shift"] = mean_shift(g, c, mask)
            run[key] = r
        # the reference against itself: seed A against seed B, the same estimator (what "no bias" looks like at this spp)
        run["reference_seed_a_vs_b"] = {"mean_shift": mean_shift(ca[..., :3].astype(np.float64), cb[..., :3].astype(np.float64), mask)}
        out["runs"].append(run)
        print(f"[parity_vs_spp] {cfg} {spp} spp: ratio {run['fast_bsdf']['ratio_to_floor']:.3f} (exact {run['exact_bsdf']['ratio_to_floor']:.3f}), luminance shift "
              f"{run['fast_bsdf']['mean_shift']['luminance']['rel_shift']:+.2e} +- {run['fast_bsdf']['mean_shift']['luminance']['se']:.1e}", file=sys.stderr, flush=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
