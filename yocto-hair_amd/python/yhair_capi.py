"""ctypes binding of include/yhair.h (the product's C ABI, libyhair.so).

Plumbing only: no arithmetic of the hot path happens in Python. The same
structures are handed to the CPU oracle (oracle/libyh_oracle.so) by the tests,
which is why the struct mirrors live here and the oracle loader lives in
tests/ (tests/oracle_capi.py) — the product never imports the oracle.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get("YHAIR_LIB", os.path.join(ROOT, "yocto-hair_amd", "libyhair.so"))

YH_OK, YH_E_INVALID, YH_E_DEVICE, YH_E_STATE, YH_E_IO, YH_E_SELFTEST = 0, -1, -2, -3, -4, -5
YH_HAIR_BRDF_FLOATS = 30
SURFACE_BSDF_FLOATS = 29
(LOBE_DIFFUSE, LOBE_SPECULAR, LOBE_METAL, LOBE_TRANSMISSION, LOBE_REFRACTION, LOBE_DELTA_SPECULAR,
 LOBE_DELTA_METAL, LOBE_DELTA_TRANSMISSION, LOBE_DELTA_REFRACTION) = range(9)
LOBE_COUNT = 9

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class Shape(C.Structure):
    _fields_ = [("num_vertices", C.c_int), ("positions", c_float_p), ("normals", c_float_p),
                ("radius", c_float_p), ("num_lines", C.c_int), ("lines", c_int_p),
                ("num_triangles", C.c_int), ("triangles", c_int_p), ("texcoords", c_float_p)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("is_byte", C.c_int), ("pixels", C.c_void_p)]


class Material(C.Structure):
    _fields_ = [("emission", C.c_float * 3), ("color", C.c_float * 3), ("specular", C.c_float),
                ("metallic", C.c_float), ("roughness", C.c_float), ("transmission", C.c_float),
                ("opacity", C.c_float), ("ior", C.c_float), ("thin", C.c_int),
                ("sigma_a", C.c_float * 3), ("beta_m", C.c_float), ("beta_n", C.c_float),
                ("alpha", C.c_float), ("eta", C.c_float), ("eumelanin", C.c_float),
                ("pheomelanin", C.c_float), ("scattering", C.c_float * 3), ("scanisotropy", C.c_float),
                ("trdepth", C.c_float), ("emission_tex", C.c_int), ("color_tex", C.c_int),
                ("scattering_tex", C.c_int)]


class Object(C.Structure):
    _fields_ = [("frame", C.c_float * 12), ("shape", C.c_int), ("material", C.c_int)]


class Environment(C.Structure):
    _fields_ = [("frame", C.c_float * 12), ("emission", C.c_float * 3), ("tex_width", C.c_int),
                ("tex_height", C.c_int), ("texels", c_float_p)]


class Camera(C.Structure):
    _fields_ = [("frame", C.c_float * 12), ("lens", C.c_float), ("film", C.c_float * 2),
                ("focus", C.c_float), ("aperture", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("num_shapes", C.c_int), ("shapes", C.POINTER(Shape)),
                ("num_materials", C.c_int), ("materials", C.POINTER(Material)),
                ("num_objects", C.c_int), ("objects", C.POINTER(Object)),
                ("num_environments", C.c_int), ("environments", C.POINTER(Environment)),
                ("camera", Camera), ("num_textures", C.c_int), ("textures", C.POINTER(Texture))]


SHADER_NAMES = ["naive", "path", "eyelight", "normal"]  # shader_type, yocto_pathtrace.h:177-199


class TraceParams(C.Structure):
    _fields_ = [("resolution", C.c_int), ("bounces", C.c_int), ("clamp", C.c_float),
                ("seed", C.c_uint64), ("shader", C.c_int), ("hair_exact", C.c_int)]

    @staticmethod
    def default(resolution=720, bounces=8, clamp=100.0, seed=961748941, shader="path", hair_exact=False):
        return TraceParams(resolution, bounces, clamp, seed, SHADER_NAMES.index(shader) if isinstance(shader, str) else shader,
                           1 if hair_exact else 0)


class WorkCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "nodes", "seg_tests", "tri_tests",
                                          "hair_shades", "surf_shades", "env_lookups",
                                          "env_samples", "cyc_trace", "cyc_shade", "ticks_tile",
                                          "wave_iters", "wave_steps", "lane_steps", "lane_iters", "cyc_geom", "cyc_sample",
                                          "cyc_eval", "cyc_rest")] + [("branch", C.c_uint64 * 10)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "branch"}
        names = ("node", "line", "tri", "enter", "scene")
        for k, nm in enumerate(names):
            d["trips_" + nm], d["lanes_" + nm] = int(self.branch[2 * k]), int(self.branch[2 * k + 1])
        return d

    def bytes_per_sample(self, spp_per_launch, env_textured=True):
        """algorithmic bytes per sample of the reference algorithm. The environment's texels count for a
        TEXTURED environment only (the oracle counts every eval_environment, yh_oracle.cpp; a constant one reads nothing)."""
        s = max(1, self.samples)
        return (32 * self.nodes + 44 * self.seg_tests + 52 * self.tri_tests +
                104 * self.hair_shades + (48 * self.env_lookups if env_textured else 0) + 88 * self.env_samples) / s \
            + 32.0 / spp_per_launch


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


def hair_material_rows(mats12):
    """(n,12) float rows [sigma_a3 beta_m beta_n alpha eta color3 eumelanin pheomelanin]
    (yocto_extension.h:86-95 order) -> ctypes array of yh_material."""
    mats12 = np.ascontiguousarray(mats12, dtype=np.float32).reshape(-1, 12)
    arr = (Material * len(mats12))()
    for i, r in enumerate(mats12):
        m = arr[i]
        m.sigma_a[:] = r[0:3].tolist()
        m.beta_m, m.beta_n, m.alpha, m.eta = map(float, r[3:7])
        m.color[:] = r[7:10].tolist()
        m.eumelanin, m.pheomelanin = float(r[10]), float(r[11])
        m.opacity, m.ior, m.thin, m.trdepth = 1.0, 1.5, 1, 0.01
    return arr


_SIGS = {
    "yh_create": (C.c_void_p, [C.c_int]),
    "yh_destroy": (None, [C.c_void_p]),
    "yh_last_error": (C.c_char_p, [C.c_void_p]),
    "yh_version": (C.c_char_p, []),
    "yh_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "yh_init_state": (C.c_int, [C.c_void_p, C.POINTER(TraceParams)]),
    "yh_image_size": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "yh_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "yh_trace_samples": (C.c_int, [C.c_void_p, C.c_int]),
    "yh_trace_samples_async": (C.c_int, [C.c_void_p, C.c_int]),
    "yh_synchronize": (C.c_int, [C.c_void_p]),
    "yh_download": (C.c_int, [C.c_void_p, c_float_p]),
    "yh_pack_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "yh_unpack_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "yh_gather_framebuffer": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_float_p]),
    "yh_shard_pixels": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "yh_download_rng": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "yh_trace_samples_counted": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(WorkCounts)]),
    "yh_last_trace_ms": (C.c_int, [C.c_void_p, c_float_p, c_int_p]),
    "yh_launch_shape": (C.c_int, [C.c_void_p]),
    "yh_tile_costs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "yh_item_costs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "yh_kernel_trials": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    "yh_trials_pending": (C.c_int, [C.c_void_p]),
    "yh_set_trial_cache_dir": (C.c_int, [C.c_char_p]),
    "yh_default_trial_cache_dir": (C.c_char_p, []),
    "yh_hair_brdf_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Material), c_float_p,
                                     c_float_p, c_float_p, c_float_p]),
    "yh_curves_to_lines": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, C.c_int, c_float_p,
                                     c_float_p, c_float_p, c_int_p]),
    "yh_bvh_build_gpu": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_int_p]),
    "yh_bvh_build": (C.c_int, [C.c_int, c_float_p, c_float_p, c_int_p]),
    "yh_bvh_build_wide": (C.c_int, [C.c_int, c_float_p, C.c_int, c_float_p]),
    "yh_bvh_build_wide_gpu": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_int, c_float_p]),
    "yh_surface_lobe_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p,
                                        c_float_p, c_float_p]),
    "yh_surface_bsdf_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Material), c_float_p, c_float_p, c_float_p,
                                        c_float_p, c_float_p]),
    "yh_hair_eval_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "yh_hair_sample_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "yh_hair_pdf_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "yh_hair_eval_pdf_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "yh_intersect_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_int_p, c_int_p, c_float_p, c_float_p]),
    "yh_selftest": (C.c_int, [C.c_void_p, C.c_int, c_float_p]),
    "yh_scene_load": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "yh_scene_get": (C.POINTER(SceneDesc), [C.c_void_p]),
    "yh_scene_free": (None, [C.c_void_p]),
    "yh_save_image": (C.c_int, [C.c_char_p, C.c_int, C.c_int, c_float_p, C.c_char_p, C.c_int]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load(path=None):
    """Loads libyhair.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None or path:
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the product path)")
        lib = C.CDLL(p)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if "YHAIR_LIB" in os.environ:  # developer A/B against an older build (tools/_ab/): it may lack newer entry points
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        if path:
            return lib
        _lib = lib
    return _lib


class YhError(RuntimeError):
    pass


def set_trial_cache_dir(path=None, default=False):
    """yh_set_trial_cache_dir: opt in to the kernel-trial record on disk (process-wide). default=True: the directory the
    command lines use ($XDG_CACHE_HOME/yhair or ~/.cache/yhair); path=None without default: no file."""
    lib = load()
    if not hasattr(lib, "yh_set_trial_cache_dir"):  # an older developer build (YHAIR_LIB)
        return None
    d = lib.yh_default_trial_cache_dir() if default else (path.encode() if path else None)
    lib.yh_set_trial_cache_dir(d)
    return d.decode() if d else None


class SceneFile:
    """yh_scene_load / yh_scene_get / yh_scene_free."""

    def __init__(self, json_path, camera=""):
        lib = load()
        err = C.create_string_buffer(512)
        self.handle = lib.yh_scene_load(str(json_path).encode(), camera.encode(), err, 512)
        if not self.handle:
            raise YhError(err.value.decode())
        self.desc = lib.yh_scene_get(self.handle)

    def close(self):
        if self.handle and load is not None:
            load().yh_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


def gather_framebuffer(contexts):
    """yh_gather_framebuffer: the full (H, W, 4) image of `contexts`, context i holding shard (i, len(contexts))."""
    lib = load()
    w, h = C.c_int(), C.c_int()
    contexts[0]._chk(lib.yh_image_size(contexts[0].h, C.byref(w), C.byref(h)))
    img = np.zeros((h.value, w.value, 4), np.float32)
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    contexts[0]._chk(lib.yh_gather_framebuffer(arr, len(contexts), fptr(img)))
    return img


class Context:
    """One GPU context (yh_create ... yh_destroy). Raises when no GPU: no fallback."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = self.lib.yh_create(device)
        if not self.h:
            raise YhError("yh_create failed: " + self.lib.yh_last_error(None).decode())

    def _chk(self, rc):
        if rc != YH_OK:
            raise YhError(f"yhair error {rc}: " + self.lib.yh_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.yh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # whole path -----------------------------------------------------------
    def upload_scene(self, desc):
        self._chk(self.lib.yh_upload_scene(self.h, desc))

    def set_shard(self, rank, world):
        self._chk(self.lib.yh_set_shard(self.h, rank, world))

    def init_state(self, params):
        self._chk(self.lib.yh_init_state(self.h, C.byref(params)))
        w, h = C.c_int(), C.c_int()
        self._chk(self.lib.yh_image_size(self.h, C.byref(w), C.byref(h)))
        self.width, self.height = w.value, h.value
        return self.width, self.height

    def trace_samples(self, n):
        self._chk(self.lib.yh_trace_samples(self.h, n))

    def trace_samples_async(self, n):
        self._chk(self.lib.yh_trace_samples_async(self.h, n))

    def synchronize(self):
        self._chk(self.lib.yh_synchronize(self.h))

    def trace_samples_counted(self, n):
        wc = WorkCounts()
        self._chk(self.lib.yh_trace_samples_counted(self.h, n, C.byref(wc)))
        return wc

    def launch_shape(self):
        """The sample-loop kernel of the most recent launch (include/yhair.h: yh_launch_shape)."""
        return int(self.lib.yh_launch_shape(self.h))

    def last_trace_ms(self):
        ms, n = C.c_float(), C.c_int()
        self._chk(self.lib.yh_last_trace_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def tile_costs(self):
        tx, ty = (self.width + 7) // 8, (self.height + 7) // 8
        out = np.zeros(tx * ty, np.uint32)
        self._chk(self.lib.yh_tile_costs(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), len(out)))
        return out.reshape(ty, tx)

    def kernel_trials(self):
        """{launch shape: (ms per sample of its fastest trial, number of trials)} for the shapes tried on this image."""
        ms, tr = (C.c_double * 16)(), (C.c_int * 16)()
        n = self.lib.yh_kernel_trials(self.h, ms, tr, 16)
        self._chk(min(n, 0))
        return {k: (round(ms[k], 5), tr[k]) for k in range(min(n, 16)) if tr[k] or ms[k]}

    def trials_pending(self):
        """True while a candidate kernel of this image still wants a 32-sample timing trial (include/yhair.h)."""
        n = self.lib.yh_trials_pending(self.h)
        self._chk(min(n, 0))
        return n > 0

    def item_costs(self):
        """Per work item (tile * 4 + quadrant) cost of the most recent launch."""
        w, h = C.c_int(), C.c_int()
        self._chk(self.lib.yh_image_size(self.h, C.byref(w), C.byref(h)))
        n = ((w.value + 7) // 8) * ((h.value + 7) // 8) * 4
        out = np.zeros(n, np.uint32)
        self._chk(self.lib.yh_item_costs(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def download(self):
        img = np.zeros((self.height, self.width, 4), np.float32)
        self._chk(self.lib.yh_download(self.h, fptr(img)))
        return img

    def download_rng(self):
        rng = np.zeros((self.height * self.width, 2), np.uint64)
        self._chk(self.lib.yh_download_rng(self.h, rng.ctypes.data_as(C.POINTER(C.c_uint64))))
        return rng

    def shard_pixels(self, rank, world):
        return int(self.lib.yh_shard_pixels(self.h, rank, world))

    def pack_tiles_device(self, dev_ptr, capacity):
        n = C.c_int64()
        self._chk(self.lib.yh_pack_tiles_device(self.h, C.c_void_p(dev_ptr), capacity, C.byref(n)))
        return n.value

    def unpack_tiles_device(self, packed_ptr, src_rank, world, image_ptr):
        self._chk(self.lib.yh_unpack_tiles_device(self.h, C.c_void_p(packed_ptr), src_rank, world,
                                                  C.c_void_p(image_ptr)))

    # unit level -----------------------------------------------------------
    def hair_brdf(self, mats12, v, normal, tangent):
        mats = hair_material_rows(mats12)
        n = len(mats)
        v = np.ascontiguousarray(v, np.float32)
        normal = np.ascontiguousarray(normal, np.float32)
        tangent = np.ascontiguousarray(tangent, np.float32)
        out = np.zeros((n, 30), np.float32)
        self._chk(self.lib.yh_hair_brdf_batch(self.h, n, mats, fptr(v), fptr(normal), fptr(tangent), fptr(out)))
        return out

    def _wowi(self, fn, brdf, a, b, width):
        brdf = np.ascontiguousarray(brdf, np.float32)
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        n = len(brdf)
        out = np.zeros((n, width) if width > 1 else (n,), np.float32)
        self._chk(fn(self.h, n, fptr(brdf), fptr(a), fptr(b), fptr(out)))
        return out

    def hair_eval(self, brdf, wo, wi):
        return self._wowi(self.lib.yh_hair_eval_batch, brdf, wo, wi, 3)

    def hair_sample(self, brdf, wo, rn):
        return self._wowi(self.lib.yh_hair_sample_batch, brdf, wo, rn, 3)

    def hair_pdf(self, brdf, wo, wi):
        return self._wowi(self.lib.yh_hair_pdf_batch, brdf, wo, wi, 1)

    def curves_to_lines(self, P, width0, width1, base_vertex=0):
        """pbrt curves (n, 12) -> positions (5n, 3), tangents (5n, 3), radius (5n), lines (4n, 2)."""
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 12)
        w0, w1 = np.ascontiguousarray(width0, np.float32), np.ascontiguousarray(width1, np.float32)
        n = len(P)
        pos, nrm = np.zeros((5 * n, 3), np.float32), np.zeros((5 * n, 3), np.float32)
        rad, lines = np.zeros(5 * n, np.float32), np.zeros((4 * n, 2), np.int32)
        self._chk(self.lib.yh_curves_to_lines(self.h, n, fptr(P), fptr(w0), fptr(w1), base_vertex, fptr(pos), fptr(nrm),
                                              fptr(rad), iptr(lines)))
        return pos, nrm, rad, lines

    def surface_lobe(self, kind, params8, normal, wo, wi, rn3):
        """One YH_LOBE_* kind (yocto_math.h:4427-4755): (n, 7) = f*|cos| [3], pdf, sampled incoming [3]."""
        a = [np.ascontiguousarray(x, np.float32) for x in (params8, normal, wo, wi, rn3)]
        out = np.zeros((len(a[0]), 7), np.float32)
        self._chk(self.lib.yh_surface_lobe_batch(self.h, kind, len(a[0]), *(fptr(x) for x in a), fptr(out)))
        return out

    def surface_bsdf(self, materials, normal, wo, wi, rn3):
        """eval_brdf + lobe dispatch of non-hair materials: (n, SURFACE_BSDF_FLOATS)."""
        a = [np.ascontiguousarray(x, np.float32) for x in (normal, wo, wi, rn3)]
        out = np.zeros((len(a[0]), SURFACE_BSDF_FLOATS), np.float32)
        self._chk(self.lib.yh_surface_bsdf_batch(self.h, len(a[0]), materials, *(fptr(x) for x in a), fptr(out)))
        return out

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = len(rays)
        obj, elem = np.zeros(n, np.int32), np.zeros(n, np.int32)
        uv, dist = np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
        self._chk(self.lib.yh_intersect_batch(self.h, n, fptr(rays), iptr(obj), iptr(elem), fptr(uv), fptr(dist)))
        return obj, elem, uv, dist

    def selftest(self, which):
        worst = C.c_float()
        rc = self.lib.yh_selftest(self.h, which, C.byref(worst))
        if rc not in (YH_OK, YH_E_SELFTEST):
            self._chk(rc)
        return rc == YH_OK, worst.value
