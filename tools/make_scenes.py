#!/usr/bin/env python3
"""Procedural stand-ins for the four hair models the reference does not ship.

`hair-block.ply`, `straight-hair.ply`, `curly-hair.ply` and `hair-curl.ply` are
git-ignored in the reference (tests/.gitignore:7-10; they live on Google Drive,
README.md:14,70), so EVERY config of BASELINE.json runs on synthetic hair
written in the reference's PLY layout (vertex `x y z nx ny nz radius` where
nx ny nz is the curve TANGENT, element `line` with `list uchar int
vertex_indices`; libs/yocto/yocto_shape.cpp:4855-4878, yocto_ply.h:1145-1161).

The scene JSONs written here are the bench variants SURVEY.md 8(d) specifies
(aspect 1.0 where the metric wants a square image, `eumelanin 1.3` instead of
`color` on the hair block, beta_m / beta_n / alpha overrides). The same files
feed the reference (oracle/_ref), the CPU oracle and the GPU path.

Geometry is deterministic: numpy Generator(PCG64) seeded 7.
"""
import argparse
import json
import os
import shutil

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
IDENT = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]


def write_hair_ply(path, pos, r_root, r_tip):
    """pos: (S, V, 3) float strand polylines. Tangents = normalised central differences,
    radius tapered linearly root -> tip."""
    S, V, _ = pos.shape
    assert V <= 255, "list length is a uchar in the reference reader"
    tang = np.empty_like(pos)
    tang[:, 1:-1] = pos[:, 2:] - pos[:, :-2]
    tang[:, 0] = pos[:, 1] - pos[:, 0]
    tang[:, -1] = pos[:, -1] - pos[:, -2]
    tang /= np.maximum(np.linalg.norm(tang, axis=2, keepdims=True), 1e-20)
    t = np.linspace(0.0, 1.0, V, dtype=np.float32)[None, :, None]
    radius = (r_root * (1 - t) + r_tip * t).astype(np.float32) * np.ones((S, 1, 1), np.float32)
    verts = np.concatenate([pos, tang, radius], axis=2).astype("<f4").reshape(S * V, 7)
    idx = np.arange(S * V, dtype="<i4").reshape(S, V)
    lines = np.empty(S, dtype=[("n", "u1"), ("i", "<i4", (V,))])
    lines["n"] = V
    lines["i"] = idx
    header = ("ply\nformat binary_little_endian 1.0\ncomment synthetic hair (tools/make_scenes.py)\n"
              f"element vertex {S * V}\nproperty float x\nproperty float y\nproperty float z\n"
              "property float nx\nproperty float ny\nproperty float nz\nproperty float radius\n"
              f"element line {S}\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode())
        f.write(verts.tobytes())
        f.write(lines.tobytes())
    return S * (V - 1)


def gen_hair_block(strands=100_000, segments=16, seed=7):
    """strands rooted uniformly on local [-.5,.5]^2, growing along
    local +z to z=1 with a helical wobble of amplitude U(0,0.02)."""
    rng = np.random.default_rng(seed)
    root = rng.uniform(-0.5, 0.5, (strands, 2)).astype(np.float32)
    amp = rng.uniform(0.0, 0.02, (strands, 1)).astype(np.float32)
    phase = rng.uniform(0.0, 2 * np.pi, (strands, 1)).astype(np.float32)
    turns = rng.uniform(1.0, 3.0, (strands, 1)).astype(np.float32)
    t = np.linspace(0.0, 1.0, segments + 1, dtype=np.float32)[None, :]
    ang = phase + 2 * np.pi * turns * t
    x = root[:, :1] + amp * t * np.cos(ang)
    y = root[:, 1:] + amp * t * np.sin(ang)
    z = np.broadcast_to(t, x.shape)
    return np.stack([x, y, z], axis=2).astype(np.float32)


def _scalp_roots(rng, strands, radius=4.5, centre=(0.0, 12.0, 0.0)):
    """Roots on the upper/back part of a scalp sphere (object space), face side left bare."""
    pts = []
    n = 0
    while n < strands:
        v = rng.normal(size=(strands * 2, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        keep = (v[:, 1] > -0.25) & ~((v[:, 2] < -0.45) & (v[:, 1] < 0.55))
        v = v[keep]
        pts.append(v)
        n += len(v)
    d = np.concatenate(pts)[:strands].astype(np.float32)
    return d, (np.asarray(centre, np.float32) + radius * d).astype(np.float32)


def gen_head_hair(strands=50_000, segments=32, curly=False, seed=7):
    """C2 (straight, 2 lines per span) / C3 (helical, 4 lines per span): strands leave the
    scalp along its normal and bend under gravity down to y ~ 2.5."""
    rng = np.random.default_rng(seed)
    d, root = _scalp_roots(rng, strands)
    length = rng.uniform(8.5, 10.5, (strands, 1)).astype(np.float32)
    t = np.linspace(0.0, 1.0, segments + 1, dtype=np.float32)[None, :, None]
    s = t * length[:, None, :]
    # outward for ~1.2 units, then blend to straight down; slight per-strand sway
    blend = (1 - np.exp(-s / 1.2)).astype(np.float32)
    out_dir = d[:, None, :]
    down = np.array([0, -1, 0], np.float32)[None, None, :]
    sway = rng.normal(0, 0.06, (strands, 1, 3)).astype(np.float32)
    sway[:, :, 1] = 0
    # integrate direction along the strand
    direction = (1 - blend) * out_dir + blend * (down + sway)
    direction /= np.linalg.norm(direction, axis=2, keepdims=True)
    step = (length / segments)[:, None, :]
    pos = root[:, None, :] + np.concatenate(
        [np.zeros((strands, 1, 3), np.float32), np.cumsum(direction[:, :-1] * step, axis=1)], axis=1)
    if curly:
        amp = rng.uniform(0.10, 0.28, (strands, 1, 1)).astype(np.float32)
        turns = rng.uniform(5.0, 9.0, (strands, 1, 1)).astype(np.float32)
        phase = rng.uniform(0, 2 * np.pi, (strands, 1, 1)).astype(np.float32)
        ang = phase + 2 * np.pi * turns * t
        grow = np.minimum(1.0, 4 * t).astype(np.float32)  # no curl right at the root
        ex = np.array([1, 0, 0], np.float32)[None, None, :]
        ez = np.array([0, 0, 1], np.float32)[None, None, :]
        pos = pos + amp * grow * (np.cos(ang) * ex + np.sin(ang) * ez)
    return pos.astype(np.float32)


def gen_hair_curl(strands=10_000, segments=100, seed=7):
    """C4: one curl bundle following a vertical helix (axis y from ~11 down to ~1)."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 1.0, segments + 1, dtype=np.float32)[None, :]
    off_r = 0.35 * np.sqrt(rng.uniform(0, 1, (strands, 1))).astype(np.float32)
    off_a = rng.uniform(0, 2 * np.pi, (strands, 1)).astype(np.float32)
    jitter = rng.uniform(-0.15, 0.15, (strands, 1)).astype(np.float32)
    turns = 4.0
    ang = 2 * np.pi * turns * t + jitter
    helix_r = 0.55 * (0.6 + 0.4 * t)
    y = 11.0 - 10.0 * t + 0.15 * off_r * np.sin(off_a)
    x = helix_r * np.cos(ang) + off_r * np.cos(off_a + 1.5 * ang)
    z = helix_r * np.sin(ang) + off_r * np.sin(off_a + 1.5 * ang)
    return np.stack([x, np.broadcast_to(y, x.shape), z], axis=2).astype(np.float32)


def _dump(scene, out_json):
    with open(out_json, "w") as f:
        json.dump(scene, f, indent=2)


def _prep(outdir, name):
    d = os.path.join(outdir, name)
    os.makedirs(os.path.join(d, "shapes"), exist_ok=True)
    os.makedirs(os.path.join(d, "textures"), exist_ok=True)
    return d


def make_sphere_hairblock(outdir, scale=1.0, name="sphere-hairblock", zoom=False, hair=None, dof=False):
    """C0/C1: variant of tests/sphere-hairblock/sphere-hairblock.json: aspect 1.0 and hair
    material {eumelanin 1.3} (the committed `color` would override melanin, ext.cpp:131-138)."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-block.ply"),
                          gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
    cam = {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [-0.5, 1.5, 5, 0.25, 0.5, 0, 0, 1, 0]}
    if zoom:  # SURVEY.md appendix B.3 "hair fills frame"
        cam = {"lens": 0.22, "aperture": 0.0, "aspect": 1.0, "lookat": [-0.5, 1.5, 5, 0.5, 0.5, -0.5, 0, 1, 0]}
    if dof:  # thin-lens branch of sample_camera (pt.cpp:211-229) and the portrait branch of init_state (pt.cpp:1933-1939)
        cam = {"lens": 0.085, "aperture": 0.12, "aspect": 0.75, "lookat": [-0.5, 1.5, 5, 0.4, 0.5, -0.3, 0, 1, 0]}
    scene = {
        "asset": {"copyright": "synthetic hair block; sphere from the reference's test assets"},
        "cameras": {"default": cam},
        "environments": {"sky": {"emission": [1, 1, 1]}},
        "objects": {
            "sphere": {"frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, -0.5, 0, 0], "shape": "sphere", "material": "diffuse"},
            "hairblock": {"frame": [1, 0, 0, 0, 0, 1, 0, -1, 0, 0.5, 1, -0.5], "shape": "hair-block", "material": "hair"},
        },
        "materials": {
            "diffuse": {"color": [0.8, 0.4, 0.05]},
            "hair": hair if hair is not None else {"eumelanin": 1.3},
            "arealight": {"emission": [20, 20, 20]},
        },
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


LOBE_MATERIALS = {
    # one sphere per lobe combination of pt.cpp:405-471
    "plastic": {"color": [0.8, 0.2, 0.2], "specular": 1.0, "roughness": 0.2},
    "roughmetal": {"color": [0.9, 0.7, 0.3], "metallic": 1.0, "roughness": 0.3},
    "mirror": {"color": [0.9, 0.9, 0.9], "metallic": 1.0, "roughness": 0.0},
    "polish": {"color": [0.0, 0.0, 0.0], "specular": 1.0, "roughness": 0.0},
    "frosted": {"color": [0.9, 1.0, 0.9], "specular": 1.0, "transmission": 1.0, "thin": True, "roughness": 0.1},
    "pane": {"color": [1.0, 1.0, 1.0], "specular": 1.0, "transmission": 1.0, "thin": True, "roughness": 0.0, "ior": 1.33},
    "veil": {"color": [0.2, 0.8, 0.3], "opacity": 0.5},
    "mixed": {"color": [0.5, 0.6, 0.9], "specular": 0.5, "metallic": 0.3, "transmission": 0.4, "thin": True,
              "roughness": 0.15, "opacity": 0.9},
}


def make_lobes(outdir, scale=1.0, name="lobes"):
    """Rank-1 widening scene: eight spheres (specular / metal / delta / thin transmission /
    opacity materials), the hair block, one area light and a constant environment."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    shutil.copy(os.path.join(ASSETS, "arealight.ply"), os.path.join(d, "shapes", "arealight.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-block.ply"),
                          gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
    objects = {
        "hairblock": {"frame": [1, 0, 0, 0, 0, 1, 0, -1, 0, 1.6, 1, -0.5], "shape": "hair-block", "material": "hair"},
        "floor": {"frame": [2, 0, 0, 0, 0, -2, 0, 2, 0, 0.3, 0, 0], "shape": "arealight", "material": "floor"},
        "light": {"lookat": [0.3, 5, 2, 0.3, 0.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"},
    }
    materials = {"hair": {"eumelanin": 1.3}, "floor": {"color": [0.6, 0.6, 0.6]}, "arealight": {"emission": [12, 12, 12]}}
    for k, (mname, mat) in enumerate(LOBE_MATERIALS.items()):
        x, z = -1.5 + 0.85 * (k % 4), (0.45 if k < 4 else -0.55)
        sz = 0.7
        objects["ball%d" % k] = {"frame": [sz, 0, 0, 0, sz, 0, 0, 0, sz, x, 0.0, z], "shape": "sphere", "material": mname}
        materials[mname] = mat
    scene = {
        "asset": {"copyright": "synthetic; sphere and quad from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [0.2, 2.6, 5.2, 0.2, 0.35, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [0.5, 0.5, 0.5]}},
        "objects": objects,
        "materials": materials,
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


VOLUME_MATERIALS = {
    # closed transmissive objects = homogeneous volumes (pt.cpp:498-533)
    "glass": {"color": [0.92, 0.97, 0.95], "specular": 1.0, "transmission": 1.0, "thin": False, "roughness": 0.0,
              "trdepth": 0.5},
    "roughglass": {"color": [0.95, 0.85, 0.7], "specular": 1.0, "transmission": 1.0, "thin": False, "roughness": 0.15,
                   "trdepth": 0.3},
    # the sloth body / bold-man skin recipe (tests/sloth/sloth.json:90-103), texture replaced by a colour
    "skin": {"color": [0.823, 0.516, 0.257], "specular": 1.0, "transmission": 1.0, "thin": False, "roughness": 0.5,
             "scattering": [0.823, 0.516, 0.257], "scanisotropy": -0.8, "trdepth": 0.001},
    "jade": {"color": [0.3, 0.7, 0.4], "specular": 1.0, "transmission": 0.8, "thin": False, "roughness": 0.2,
             "scattering": [0.5, 0.9, 0.6], "scanisotropy": 0.4, "trdepth": 0.05, "opacity": 0.95},
}


def make_volumes(outdir, scale=1.0, name="volumes"):
    """Rank-2 widening scene: four closed spheres with a medium inside (clear and rough glass,
    skin-like and jade-like subsurface scattering), the hair block, one area light."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    shutil.copy(os.path.join(ASSETS, "arealight.ply"), os.path.join(d, "shapes", "arealight.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-block.ply"),
                          gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
    objects = {
        "hairblock": {"frame": [1, 0, 0, 0, 0, 1, 0, -1, 0, 1.6, 1, -0.5], "shape": "hair-block", "material": "hair"},
        "floor": {"frame": [2, 0, 0, 0, 0, -2, 0, 2, 0, 0.3, 0, 0], "shape": "arealight", "material": "floor"},
        "light": {"lookat": [0.3, 5, 2, 0.3, 0.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"},
    }
    materials = {"hair": {"eumelanin": 0.3}, "floor": {"color": [0.6, 0.6, 0.6]}, "arealight": {"emission": [12, 12, 12]}}
    for k, (mname, mat) in enumerate(VOLUME_MATERIALS.items()):
        objects["ball%d" % k] = {"frame": [0.8, 0, 0, 0, 0.8, 0, 0, 0, 0.8, -1.5 + 0.95 * k, 0.0, 0.3], "shape": "sphere",
                                 "material": mname}
        materials[mname] = mat
    scene = {
        "asset": {"copyright": "synthetic; sphere and quad from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [0.2, 2.2, 5.0, 0.2, 0.4, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [0.5, 0.5, 0.5]}},
        "objects": objects,
        "materials": materials,
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


def _write_uv_quad(path, half=2.0, tiles=2.0):
    """A quad in the XY plane (normal +z) with texture coordinates that run past [0, 1] (tiling)."""
    verts = [(-half, -half, 0, 0, 0, 1, 0, 0), (half, -half, 0, 0, 0, 1, tiles, 0),
             (-half, half, 0, 0, 0, 1, 0, tiles), (half, half, 0, 0, 0, 1, tiles, tiles)]
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                "property float nx\nproperty float ny\nproperty float nz\nproperty float u\nproperty float v\n"
                "element face 1\nproperty list uchar int vertex_indices\nend_header\n")
        for v in verts:
            f.write(" ".join(repr(float(x)) for x in v) + "\n")
        f.write("4 0 1 3 2\n")


def make_textured(outdir, scale=1.0, name="textured"):
    """Colour textures: the sloth scene's floor.png as color_tex on a tiled
    floor with texture coordinates, as emission_tex on a light, as scattering_tex inside a volume
    (the bold-man recipe), sky.hdr as a float colour texture on a sphere without texture coordinates."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    shutil.copy(os.path.join(ASSETS, "floor.png"), os.path.join(d, "textures", "floor.png"))
    shutil.copy(os.path.join(ASSETS, "sky.hdr"), os.path.join(d, "textures", "sky.hdr"))
    _write_uv_quad(os.path.join(d, "shapes", "uvquad.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-block.ply"),
                          gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
    objects = {
        "hairblock": {"frame": [1, 0, 0, 0, 0, 1, 0, -1, 0, 1.4, 1, -0.5], "shape": "hair-block", "material": "hair"},
        "floor": {"frame": [2, 0, 0, 0, 0, -2, 0, 2, 0, 0.3, 0, 0], "shape": "uvquad", "material": "floor"},
        "light": {"lookat": [0.3, 5, 2, 0.3, 0.5, 0, 0, 1, 0], "shape": "uvquad", "material": "panel"},
        "ball0": {"frame": [0.8, 0, 0, 0, 0.8, 0, 0, 0, 0.8, -1.3, 0.0, 0.3], "shape": "sphere", "material": "marble"},
        "ball1": {"frame": [0.8, 0, 0, 0, 0.8, 0, 0, 0, 0.8, -0.3, 0.0, 0.3], "shape": "sphere", "material": "skin"},
        "ball2": {"frame": [0.8, 0, 0, 0, 0.8, 0, 0, 0, 0.8, 0.7, 0.0, 0.3], "shape": "sphere", "material": "glazed"},
    }
    materials = {
        "hair": {"eumelanin": 0.3},
        "floor": {"color": [0.7, 0.7, 0.7], "color_tex": "floor"},
        "panel": {"emission": [14, 14, 14], "emission_tex": "floor"},
        "marble": {"color": [0.9, 0.9, 0.9], "color_tex": "sky", "specular": 1.0, "roughness": 0.1},
        "skin": {"color": [0.8, 0.8, 0.8], "color_tex": "floor", "specular": 1.0, "transmission": 1.0, "thin": False,
                 "roughness": 0.5, "scattering": [1.0, 1.0, 1.0], "scattering_tex": "floor", "scanisotropy": -0.8,
                 "trdepth": 0.001},
        "glazed": {"color": [0.9, 0.8, 0.7], "color_tex": "floor", "emission": [0.2, 0.2, 0.2], "emission_tex": "sky",
                   "transmission": 0.5, "thin": True, "roughness": 0.2, "specular": 1.0},
    }
    scene = {
        "asset": {"copyright": "synthetic; sphere, floor.png and sky.hdr from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [0.2, 2.2, 5.0, 0.2, 0.4, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [0.5, 0.5, 0.5]}},
        "objects": objects,
        "materials": materials,
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


def make_crowd(outdir, scale=1.0, name="crowd", count=70):
    """Many objects: a scene-level BVH several levels deep (the BASELINE configs have at most six
    objects) and more objects than the kernel stages in LDS, so the in-memory fallback of the scene
    level runs. `count` small spheres of four materials on a jittered grid around the hair block."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    shutil.copy(os.path.join(ASSETS, "arealight.ply"), os.path.join(d, "shapes", "arealight.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-block.ply"),
                          gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
    rng = np.random.default_rng(11)
    objects = {
        "hairblock": {"frame": [1, 0, 0, 0, 0, 1, 0, -1, 0, 0.0, 1, -0.5], "shape": "hair-block", "material": "hair"},
        "light": {"lookat": [0.3, 6, 2, 0.3, 0.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"},
    }
    mats = ["red", "gold", "glass", "veil"]
    side = int(np.ceil(np.sqrt(count)))
    for k in range(count):
        x = -3.0 + 6.0 * (k % side) / max(1, side - 1) + rng.uniform(-0.1, 0.1)
        z = -2.5 + 5.0 * (k // side) / max(1, side - 1) + rng.uniform(-0.1, 0.1)
        sz = float(rng.uniform(0.15, 0.3))
        a = float(rng.uniform(0, 2 * np.pi))
        c, sn = float(np.cos(a)) * sz, float(np.sin(a)) * sz  # rotated about y and scaled: a non-trivial inverse frame
        objects["ball%03d" % k] = {"frame": [c, 0, -sn, 0, sz, 0, sn, 0, c, float(x), 0.0, float(z)], "shape": "sphere",
                                   "material": mats[k % len(mats)]}
    materials = {"hair": {"eumelanin": 1.3}, "arealight": {"emission": [15, 15, 15]},
                 "red": {"color": [0.8, 0.2, 0.2]}, "gold": {"color": [0.9, 0.7, 0.3], "metallic": 1.0, "roughness": 0.25},
                 "glass": {"color": [0.95, 0.97, 1.0], "specular": 1.0, "transmission": 1.0, "thin": False, "roughness": 0.0,
                           "trdepth": 0.3},
                 "veil": {"color": [0.2, 0.7, 0.3], "opacity": 0.6}}
    scene = {
        "asset": {"copyright": "synthetic; sphere and quad from the reference's test assets"},
        "cameras": {"default": {"lens": 0.035, "aperture": 0.0, "aspect": 1.0, "lookat": [0.5, 3.5, 7.0, 0.0, 0.3, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [0.6, 0.6, 0.6]}},
        "objects": objects,
        "materials": materials,
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


def _head_scene(outdir, name, shape, pos, emission, lights, hair_mat):
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sky.hdr"), os.path.join(d, "textures", "sky.hdr"))
    nseg = write_hair_ply(os.path.join(d, "shapes", shape + ".ply"), pos, 0.006, 0.003)
    objects = {"hair": {"frame": [-1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0], "shape": shape, "material": "brown"}}
    materials = {"brown": hair_mat}
    if lights:
        shutil.copy(os.path.join(ASSETS, "arealight_straight.ply"), os.path.join(d, "shapes", "arealight.ply"))
        objects["arealight1"] = {"lookat": [-5, 20, 10, 0, 9.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"}
        objects["arealight2"] = {"lookat": [5, 20, 10, 0, 9.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"}
        materials["arealight"] = {"emission": [20, 20, 20]}
    scene = {
        "asset": {"copyright": "synthetic head of hair; sky.hdr from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [0, 15, 23, 0, 9.5, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": emission, "emission_tex": "sky", "frame": IDENT}},
        "objects": objects,
        "materials": materials,
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg


def make_straight_hair(outdir, scale=1.0, beta_m=0.3, name=None):
    """C2: variant of tests/straight-hair/straight-hair.json with a beta_m override."""
    name = name or ("straight-hair" if beta_m == 0.3 else f"straight-hair-bm{beta_m:g}")
    mat = {"eumelanin": 1.3}
    if beta_m != 0.3:
        mat["beta_m"] = beta_m
    return _head_scene(outdir, name, "straight-hair", gen_head_hair(max(64, int(50_000 * scale)), 32, False),
                       [1, 1, 1], True, mat)


def make_curly_hair(outdir, scale=1.0, name="curly-hair"):
    """C3: tests/curly-hair/curly-hair.json (environment only, emission 2.5)."""
    return _head_scene(outdir, name, "curly-hair", gen_head_hair(max(64, int(50_000 * scale)), 64, True),
                       [2.5, 2.5, 2.5], False, {"eumelanin": 1.3})


def make_hair_curls(outdir, scale=1.0, name="hair-curls", beta_n=0.9, alpha=2):
    """C4: variant of tests/hair-curls/hair-curls.json: aspect 1.0, beta_n 0.9, alpha 2 on the
    four hair materials; one curl shape instanced four times."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sky.hdr"), os.path.join(d, "textures", "sky.hdr"))
    shutil.copy(os.path.join(ASSETS, "arealight.ply"), os.path.join(d, "shapes", "arealight.ply"))
    nseg = write_hair_ply(os.path.join(d, "shapes", "hair-curl.ply"),
                          gen_hair_curl(max(64, int(10_000 * scale))), 0.008, 0.004)
    ov = {"beta_n": beta_n, "alpha": alpha}
    objects = {}
    for nm, x in (("black", -3.75), ("red", -1.25), ("brown", 1.25), ("blonde", 3.75)):
        objects[nm] = {"frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, x, 0, 0], "shape": "hair-curl", "material": nm}
    objects["arealight1"] = {"lookat": [-8, 10, 5, -6.5, 3.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"}
    objects["arealight2"] = {"lookat": [0, 10, 5, -1.5, 3.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"}
    scene = {
        "asset": {"copyright": "synthetic curl bundle; sky.hdr / arealight.ply from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [-4, 5.9, 20, -4, 5.9, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [2, 2, 2], "emission_tex": "sky", "frame": IDENT}},
        "objects": objects,
        "materials": {
            "black": dict(eumelanin=8, **ov), "red": dict(pheomelanin=2, **ov),
            "brown": dict(eumelanin=1.3, **ov), "blonde": dict(eumelanin=0.3, **ov),
            "arealight": {"emission": [20, 20, 20]},
        },
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), nseg * 4


# ---------------------------------------------------------------------------------------------
# The reference's OWN scene descriptions (tests/golden/ref_scenes/*.json: data files of the
# reference's test directory, verbatim) with stand-ins for the models it does not distribute
# (tests/.gitignore there): synthetic hair by shape name, the reference's sphere for every mesh,
# a uv quad for the floor; textures it does ship (sky.hdr, floor.png) under their own names.
# ---------------------------------------------------------------------------------------------
REF_SCENES = os.path.join(ROOT, "tests", "golden", "ref_scenes")


def make_reference_scene(outdir, scale=1.0, name=None, which="sloth"):
    d = _prep(outdir, name)
    with open(os.path.join(REF_SCENES, which + ".json")) as f:
        scene = json.load(f)
    shapes, textures = set(), set()
    for o in scene.get("objects", {}).values():
        if "shape" in o:
            shapes.add(o["shape"])
    for group in ("materials", "environments"):
        for m in scene.get(group, {}).values():
            textures.update(v for k, v in m.items() if k.endswith("_tex"))
    nseg = 0
    for sh in sorted(shapes):
        dst = os.path.join(d, "shapes", sh + ".ply")
        if sh == "hair-block":
            nseg += write_hair_ply(dst, gen_hair_block(max(64, int(100_000 * scale))), 0.004, 0.001)
        elif sh == "straight-hair":
            nseg += write_hair_ply(dst, gen_head_hair(max(64, int(50_000 * scale)), 32, False), 0.006, 0.003)
        elif sh == "curly-hair":
            nseg += write_hair_ply(dst, gen_head_hair(max(64, int(50_000 * scale)), 64, True), 0.006, 0.003)
        elif sh == "hair-curl":
            nseg += write_hair_ply(dst, gen_hair_curl(max(64, int(10_000 * scale))), 0.008, 0.004)
        elif "hair" in sh:  # sloth's fur patches
            nseg += write_hair_ply(dst, gen_hair_curl(max(64, int(4_000 * scale)), 40), 0.008, 0.004)
        elif sh == "arealight":
            shutil.copy(os.path.join(ASSETS, "arealight_straight.ply" if which == "straight-hair" else "arealight.ply"), dst)
        elif sh == "floor":
            _write_uv_quad(dst, half=12.0, tiles=6.0)
        else:
            shutil.copy(os.path.join(ASSETS, "sphere.ply"), dst)
    for t in sorted(textures):
        if t in ("floor", "texture1"):
            shutil.copy(os.path.join(ASSETS, "floor.png"), os.path.join(d, "textures", t + ".png"))
        else:
            shutil.copy(os.path.join(ASSETS, "sky.hdr"), os.path.join(d, "textures", t + ".hdr"))
    shutil.copy(os.path.join(REF_SCENES, which + ".json"), os.path.join(d, name + ".json"))  # verbatim
    return os.path.join(d, name + ".json"), nseg


def _ref_maker(which):
    return lambda outdir, scale=1.0, name=None: make_reference_scene(outdir, scale, name or ("ref-" + which), which)


def _write_grid_light(path, n=3):
    """The reference's arealight quad (-2..2 in x and y, z = 0, facing +z) as an n x n grid: 2 n^2 triangles — an area
    light too big for the kernels' LDS light table (> 4 triangles), i.e. one sampled and intersected through memory."""
    verts = [(-2 + 4 * i / n, -2 + 4 * j / n, 0.0) for j in range(n + 1) for i in range(n + 1)]
    faces = []
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i, (j + 1) * (n + 1) + i + 1
            faces += [(a, b, c), (d, c, b)]
    with open(path, "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\nproperty float z\n"
                f"property float nx\nproperty float ny\nproperty float nz\nelement face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        for v in verts:
            f.write(f"{v[0]:.9g} {v[1]:.9g} {v[2]:.9g} 0 0 1\n")
        for t in faces:
            f.write(f"3 {t[0]} {t[1]} {t[2]}\n")


def make_lights_unit(outdir, scale=1.0, name="lights-unit", biglight=False):
    """Light sampling in isolation (pt.cpp:1283-1358): diffuse spheres and a floor — no hair, so no
    libm-driven path divergence ahead of the light code — under two area lights (the uniform light
    pick, triangle CDF, the 100-step area pdf walk through both quads) and the textured sky (texel CDF
    upper_bound, texel pdf)."""
    d = _prep(outdir, name)
    shutil.copy(os.path.join(ASSETS, "sphere.ply"), os.path.join(d, "shapes", "sphere.ply"))
    shutil.copy(os.path.join(ASSETS, "arealight.ply"), os.path.join(d, "shapes", "arealight.ply"))
    shutil.copy(os.path.join(ASSETS, "sky.hdr"), os.path.join(d, "textures", "sky.hdr"))
    objects = {
        "floor": {"frame": [3, 0, 0, 0, 0, -3, 0, 3, 0, 0.0, 0, 0], "shape": "arealight", "material": "floor"},
        "ball0": {"frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, -0.7, 0.0, 0.2], "shape": "sphere", "material": "red"},
        "ball1": {"frame": [0.6, 0, 0, 0, 0.6, 0, 0, 0, 0.6, 0.8, 0.0, 0.6], "shape": "sphere", "material": "white"},
        # one light behind the other as seen from the floor: the area pdf walk crosses both quads
        "light1": {"lookat": [0.5, 4, 1.5, 0.0, 0.5, 0, 0, 1, 0], "shape": "arealight", "material": "arealight"},
        "light2": {"lookat": [1.0, 8, 3.0, 0.0, 0.5, 0, 0, 1, 0], "shape": "gridlight" if biglight else "arealight", "material": "arealight2"},
    }
    if biglight:  # the second light as an 18-triangle mesh: the light code that goes through the BVH (GENERAL kernel variants)
        _write_grid_light(os.path.join(d, "shapes", "gridlight.ply"))
    scene = {
        "asset": {"copyright": "synthetic; sphere, quad and sky.hdr from the reference's test assets"},
        "cameras": {"default": {"lens": 0.05, "aperture": 0.0, "aspect": 1.0, "lookat": [0.0, 2.4, 5.5, 0.0, 0.5, 0, 0, 1, 0]}},
        "environments": {"sky": {"emission": [1.5, 1.5, 1.5], "emission_tex": "sky", "frame": IDENT}},
        "objects": objects,
        "materials": {"floor": {"color": [0.7, 0.7, 0.7]}, "red": {"color": [0.8, 0.2, 0.2]}, "white": {"color": [0.9, 0.9, 0.9]},
                      "arealight": {"emission": [10, 10, 10]}, "arealight2": {"emission": [30, 25, 20]}},
    }
    _dump(scene, os.path.join(d, name + ".json"))
    return os.path.join(d, name + ".json"), 0


MAKERS = {
    "lights-unit": make_lights_unit,
    "sphere-hairblock": make_sphere_hairblock,
    "straight-hair": make_straight_hair,
    "curly-hair": make_curly_hair,
    "hair-curls": make_hair_curls,
    "lobes": make_lobes,
    "volumes": make_volumes,
    "textured": make_textured,
    "crowd": make_crowd,
}


for _w in ("sloth", "bold-man", "straight-hair", "curly-hair", "hair-curls", "sphere-hairblock"):
    MAKERS["ref-" + _w] = _ref_maker(_w)


def ensure_scene(name, outdir, scale=1.0, **kw):
    """Builds the scene once per (name, scale, overrides) under outdir and returns the JSON path."""
    tag = name if scale == 1.0 else f"{name}-x{scale:g}"
    for k, v in sorted(kw.items()):
        tag += f"-{k}{v:g}" if isinstance(v, (int, float)) and not isinstance(v, bool) else (f"-{k}" if v else "")
    path = os.path.join(outdir, tag, tag + ".json")
    if not os.path.exists(path):
        MAKERS[name](outdir, scale=scale, name=tag, **kw)
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "scenes"))
    ap.add_argument("--scale", type=float, default=1.0, help="strand-count multiplier")
    ap.add_argument("scenes", nargs="*", default=list(MAKERS))
    a = ap.parse_args()
    for s in a.scenes:
        p = ensure_scene(s, a.out, a.scale)
        print(p)
