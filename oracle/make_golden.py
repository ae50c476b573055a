#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libyh_ref.so, built by
`make -C oracle ref` from /root/reference). Run in the build container only; the vectors are
committed so that the oracle can be pinned where the reference does not exist (the GPU box).

Test infrastructure: inputs and expected outputs only — no reference source is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_scenes  # noqa: E402
import oracle_capi as oc  # noqa: E402

yh = oc.yh
REF_SCENE_NAMES = ["sloth", "bold-man", "straight-hair", "curly-hair", "hair-curls", "sphere-hairblock"]
# scenes the naive / eyelight / normal shader fixtures are rendered on (name, variant, resolution)
SHADER_SCENES = [
    ("sphere-hairblock", dict(scale=0.05, zoom=True), 40),
    ("hair-curls", dict(scale=0.05), 40),
    ("lobes", dict(scale=0.05), 48),
    ("textured", dict(scale=0.05), 48),
]
GOLD = os.path.join(ROOT, "tests", "golden")

# The small scene variants every image / closest-hit fixture is rendered on. Geometry comes
# from tools/make_scenes.py (deterministic), so tests regenerate the same files.
GOLDEN_SCENES = [
    ("sphere-hairblock", dict(scale=0.02), 64),
    ("sphere-hairblock", dict(scale=0.05, zoom=True), 64),
    ("straight-hair", dict(scale=0.05), 64),
    ("straight-hair", dict(scale=0.05, beta_m=0.1), 64),
    ("curly-hair", dict(scale=0.05), 64),
    ("hair-curls", dict(scale=0.05), 64),
    ("lobes", dict(scale=0.05), 96),      # specular / metal / delta / transmission / opacity
    ("volumes", dict(scale=0.05), 96),    # refraction into homogeneous media, subsurface walk
    ("sphere-hairblock", dict(scale=0.05, dof=True), 96),  # thin lens (aperture > 0), portrait film
    ("crowd", dict(scale=0.05), 96),      # 72 objects: deep scene-level BVH, scene table larger than its LDS stage
    ("textured", dict(scale=0.05), 96),   # colour textures: color_tex / emission_tex / scattering_tex, png + hdr, tiling
    # C2's beta_m sweep {0.1, 0.25, 0.6}: 0.1 and 0.25 take Mp's v <= 0.1 branch (ext.cpp:201-207), 0.6 the other
    ("straight-hair", dict(scale=0.05, beta_m=0.25), 64),
    ("straight-hair", dict(scale=0.05, beta_m=0.6), 64),
]


def scene_tag(name, kw):
    return os.path.basename(os.path.dirname(make_scenes.ensure_scene(name, "/tmp/yhair_golden_scenes", **kw)))


def unit_dirs(rng, k):
    x = rng.normal(size=(k, 3))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def hair_inputs(rng, n):
    mats = np.zeros((n, 12), np.float32)
    mats[:, 3] = rng.uniform(0.05, 0.95, n)
    mats[:, 4] = rng.uniform(0.05, 0.95, n)
    mats[:, 5] = rng.uniform(0, 5, n)
    mats[:, 6] = rng.uniform(1.3, 1.7, n)
    mats[:, 10] = rng.uniform(0, 8, n)
    mats[:, 11] = rng.uniform(0, 2, n) * (rng.uniform(size=n) < 0.3)
    k = n // 4
    mats[:k, 7:10] = rng.uniform(0.02, 0.98, (k, 3))            # colour-parameterised rows
    mats[k:2 * k, 0:3] = rng.uniform(0.0, 2.0, (k, 3))          # explicit sigma_a rows
    # the survey's sanity rows
    mats[-1] = [0, 0, 0, 0.3, 0.3, 2, 1.55, 0, 0, 0, 1.3, 0]
    mats[-2] = [0, 0, 0, 0.3, 0.3, 2, 1.55, 0.8, 0.4, 0.05, 0, 0]
    v = rng.uniform(0, 1, n).astype(np.float32)
    v[:8] = [0, 1, 0.5, 1e-7, 0.9999999, 0.25, 0.75, 0.7]
    tng, wo, wi = unit_dirs(rng, n), unit_dirs(rng, n), unit_dirs(rng, n)
    nrm = wo - tng * np.sum(wo * tng, axis=1, keepdims=True)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    rn = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    rn[:4] = [[0, 0], [0.999999, 0.999999], [0.37, 0.81], [0.5, 0.5]]
    v[-1] = 0.7
    tng[-1] = np.array([1, .2, .1], np.float32) / np.linalg.norm([1, .2, .1])
    wo[-1] = np.array([.3, .5, .8], np.float32) / np.linalg.norm([.3, .5, .8])
    nrm[-1] = wo[-1] - tng[-1] * np.dot(wo[-1], tng[-1])
    nrm[-1] /= np.linalg.norm(nrm[-1])
    wi[-1] = np.array([-.4, .1, .6], np.float32) / np.linalg.norm([-.4, .1, .6])
    rn[-1] = [0.37, 0.81]
    return mats, v, nrm.astype(np.float32), tng, wo, wi, rn


def lobe_inputs(rng, n):
    """Inputs of the surface lobes (yocto_math.h:4427-4755): params = ior, roughness, eta[3], etak[3]."""
    nn, wo, wi = unit_dirs(rng, n), unit_dirs(rng, n), unit_dirs(rng, n)
    k = n // 4  # a quarter of the rows: incoming = what the lobe itself would reflect / transmit (the peak)
    wi[:k] = (-wo[:k] + 2 * np.sum(nn[:k] * wo[:k], 1, keepdims=True) * nn[:k] + rng.normal(0, 0.05, (k, 3))).astype(np.float32)
    wi[k:2 * k] = (-wo[k:2 * k] + rng.normal(0, 0.05, (k, 3))).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    p = np.zeros((n, 8), np.float32)
    p[:, 0] = rng.choice([1.0, 1.0005, 1.33, 1.5, 2.4], n)
    p[:, 1] = rng.choice([0.0009, 0.01, 0.04, 0.25, 1.0], n)      # brdf.roughness (already squared, pt.cpp:441)
    p[:, 2:5] = rng.uniform(0.1, 3.5, (n, 3))
    p[:, 5:8] = rng.choice([0, 1], n)[:, None] * rng.uniform(0, 4, (n, 3))
    rn = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    rn[:4] = [[0, 0, 0], [0.999999, 0.999999, 0.999999], [0.5, 0.37, 0.81], [0.5, 0.5, 0.5]]
    return p, nn, wo, wi.astype(np.float32), rn


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = oc.Ref()
    only = sys.argv[1:]  # e.g. `make_golden.py lobes` regenerates just the fixtures whose name contains "lobes"
    want = lambda tag: not only or any(o in tag for o in only)  # noqa: E731

    # ---- surface lobes (math.h:4215-4755), its own generator so older fixtures never move ------
    if want("lobes.npz"):
        lrng = np.random.default_rng(20240608)
        p, nn, wo, wi, rn = lobe_inputs(lrng, 4096)
        out = dict(params=p, normal=nn, wo=wo, wi=wi, rn=rn, fresnel=ref.fresnel(p, nn, wo))
        for kind in range(yh.LOBE_COUNT):
            out[f"lobe_{kind}"] = ref.surface_lobe(kind, p, nn, wo, wi, rn)
        np.savez_compressed(os.path.join(GOLD, "lobes.npz"), **out)
    # ---- pbrt curve -> line strands (yocto_pbrt.h:1751-1797) ------------------------------------
    if want("curves.npz"):
        crng = np.random.default_rng(20240610)
        n = 2048
        root = crng.uniform(-1, 1, (n, 1, 3))
        P = (root + np.cumsum(crng.normal(0, 0.05, (n, 4, 3)), axis=1)).astype(np.float32).reshape(n, 12)
        P[0] = [0, 0, 0, 0, 1, 0, 0, 2, 0, 0, 3, 0]                  # straight
        P[1] = [0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0]                  # p1 == p0, p3 == p2: zero end tangents
        w0 = crng.uniform(0.001, 0.01, n).astype(np.float32)
        w1 = (w0 * crng.uniform(0.1, 1.0, n)).astype(np.float32)
        pos, nrm, rad, lines = ref.curves_to_lines(P, w0, w1, 100)
        np.savez_compressed(os.path.join(GOLD, "curves.npz"), P=P, width0=w0, width1=w1, base_vertex=100,
                            positions=pos, normals=nrm, radius=rad, lines=lines)
    # ---- the other shaders: trace_naive / trace_eyelight / trace_normal (pt.cpp:1514-1672) --------
    if want("shaders.npz"):
        out = {}
        for name, kw, res in SHADER_SCENES:
            path = make_scenes.ensure_scene(name, "/tmp/yhair_golden_scenes", **kw)
            tag = os.path.basename(os.path.dirname(path))
            sc = ref.scene(path)
            for shader in ("naive", "eyelight", "normal"):
                p = yh.TraceParams.default(resolution=res, shader=shader)
                out[f"{tag}|{shader}|1"] = sc.render(p, 1)
                out[f"{tag}|{shader}|8"], out[f"{tag}|{shader}|rng8"] = sc.render(p, 8, want_rng=True)
            sc.close()
        np.savez_compressed(os.path.join(GOLD, "shaders.npz"), **out)
    # ---- the reference's own scene files, loaded by the reference's own loader -------------------
    if want("refscenes.npz"):
        out = {}
        for which in REF_SCENE_NAMES:
            path = make_scenes.ensure_scene("ref-" + which, "/tmp/yhair_golden_scenes", scale=0.05)
            sc = ref.scene(path)
            p = yh.TraceParams.default(resolution=48)
            out[f"{which}|1"] = sc.render(p, 1)
            out[f"{which}|8"], out[f"{which}|rng8"] = sc.render(p, 8, want_rng=True)
            out[f"{which}|8_seed777"] = sc.render(yh.TraceParams.default(resolution=48, seed=777), 8)
            out[f"{which}|lights"] = np.int32(sc.num_lights())
            sc.close()
        np.savez_compressed(os.path.join(GOLD, "refscenes.npz"), **out)
    rng = np.random.default_rng(20240607)
    if only:
        scenes_only(ref, rng, want)
        return

    # ---- rng (math.h:1405-1442, pt.cpp:1942-1945) -------------------------------------------
    streams = {}
    for seed, seq in ((961748941, 1), (961748941, 725124800), (12345, 7), (0, 0), (2 ** 63 + 5, 2 ** 40 + 3)):
        si, fl = ref.rng_stream(seed, seq, 64)
        streams[f"{seed}_{seq}"] = (np.array(si, np.uint64), fl)
    np.savez_compressed(os.path.join(GOLD, "rng.npz"), pixel_seqs=ref.pixel_seqs(64 * 64),
                        **{f"state_{k}": v[0] for k, v in streams.items()},
                        **{f"floats_{k}": v[1] for k, v in streams.items()})

    # ---- hair BSDF (ext.cpp:127-551) --------------------------------------------------------
    mats, v, nrm, tng, wo, wi, rn = hair_inputs(rng, 2048)
    brdf = ref.hair_brdf(mats, v, nrm, tng)
    wis = ref.hair_sample(brdf, wo, rn)
    np.savez_compressed(os.path.join(GOLD, "hair_bsdf.npz"), mats=mats, v=v, normal=nrm, tangent=tng, wo=wo, wi=wi,
                        rn=rn, brdf=brdf, f=ref.hair_eval(brdf, wo, wi), pdf=ref.hair_pdf(brdf, wo, wi),
                        wi_sampled=wis, f_sampled=ref.hair_eval(brdf, wo, wis), pdf_sampled=ref.hair_pdf(brdf, wo, wis))

    # ---- primitive tests (math.h:3426-3554) -------------------------------------------------
    n = 4096
    org = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d, np.full((n, 1), 1e-4), rng.choice([3.4028235e38, 2.0, 5.0], (n, 1))], 1).astype(np.float32)
    rays[0] = [0, 0, 1, *(np.array([.05, .02, -1]) / np.linalg.norm([.05, .02, -1])), 1e-4, 3.4028235e38]
    rays[1:9, 3:6] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0], [1, 1, 0]]
    p0 = (tgt + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
    p1 = (p0 + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
    p2 = (p0 + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
    r0 = rng.uniform(0.005, 0.3, n).astype(np.float32)
    r1 = rng.uniform(0.005, 0.3, n).astype(np.float32)
    p0[0], p1[0], r0[0], r1[0] = [-1, 0, 0], [1, .1, 0], .05, .02
    p1[9] = p0[9]                                   # degenerate segment
    p1[10] = p0[10] + rays[10, 3:6]                 # parallel to the ray (det == 0)
    bmin = np.minimum(p0, p1) - r0[:, None]
    bmax = np.maximum(p0, p1) + r1[:, None]
    bmin[11], bmax[11] = rays[11, :3], rays[11, :3] + 1  # origin on the box corner (NaN slabs)
    lh, luv, ld = ref.intersect_line(rays, p0, p1, r0, r1)
    th, tuv, td = ref.intersect_triangle(rays, p0, p1, p2)
    bh = ref.intersect_bbox(rays, np.concatenate([bmin, bmax], 1))
    np.savez_compressed(os.path.join(GOLD, "intersect.npz"), rays=rays, p0=p0, p1=p1, p2=p2, r0=r0, r1=r1,
                        bbox=np.concatenate([bmin, bmax], 1).astype(np.float32), line_hit=lh, line_uv=luv,
                        line_dist=ld, tri_hit=th, tri_uv=tuv, tri_dist=td, bbox_hit=bh)

    scenes_only(ref, rng, want)


def scenes_only(ref, rng, want):
    # ---- scenes: closest hits and images (pt.cpp:934-1046, 1380-1511, 1676-1689) -------------
    for name, kw, res in GOLDEN_SCENES:
        path = make_scenes.ensure_scene(name, "/tmp/yhair_golden_scenes", **kw)
        tag = os.path.basename(os.path.dirname(path))
        if not want("scene_" + tag):
            continue
        if name in ("lobes", "volumes", "textured", "crowd"):
            rng = np.random.default_rng(20240609)
        if kw.get("dof"):
            rng = np.random.default_rng(20240611)
        if kw.get("beta_m") in (0.25, 0.6):  # added in round 2: own generator, older fixtures never move
            rng = np.random.default_rng(20240612)
        sc = ref.scene(path)
        m = 4096
        if name in ("lobes", "volumes", "textured", "crowd"):
            org = rng.uniform(-1, 1, (m, 3)) * [3, 1.2, 1] + [0.2, 2.0, 4.5]
            tgt = rng.uniform(-1, 1, (m, 3)) * [2.4, 0.6, 1.2] + [0.3, 0.4, 0]
        elif name == "sphere-hairblock":
            org = rng.uniform(-1, 1, (m, 3)) * 2 + [0, 1.0, 4]
            tgt = rng.uniform(-1, 1, (m, 3)) * [1.2, 0.7, 0.6] + [0.25, 0.5, -0.25]
        else:
            org = rng.uniform(-1, 1, (m, 3)) * 6 + [0, 12, 20]
            tgt = rng.uniform(-1, 1, (m, 3)) * [6, 6, 3] + [-1, 8, 0]
        d = tgt - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([org, d, np.full((m, 1), 1e-4), np.full((m, 1), 3.4028235e38)], 1).astype(np.float32)
        obj, elem, uv, dist = sc.intersect(rays)
        out = dict(rays=rays, object=obj, element=elem, uv=uv, distance=dist, num_lights=sc.num_lights())
        p = yh.TraceParams.default(resolution=res)
        for spp in (1, 16):
            img, rs = sc.render(p, spp, want_rng=True)
            out[f"img_{spp}"] = img
            out[f"rng_{spp}"] = rs
        p2_ = yh.TraceParams.default(resolution=res, seed=12345)
        out["img_16_seed12345"] = sc.render(p2_, 16)
        np.savez_compressed(os.path.join(GOLD, f"scene_{tag}.npz"), **out)
        print(tag, "hits %.3f" % np.mean(obj >= 0), "mean", out["img_16"][..., :3].mean())
        sc.close()


if __name__ == "__main__":
    main()
