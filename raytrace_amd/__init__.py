"""raytrace_amd -- MI355X-native PathIntegrator back end for the Enigma renderer.

Python here is harness only: it loads the C-ABI shared library (raytrace_amd/libenigma_hip.so,
built from raytrace_amd/csrc by `make -C raytrace_amd/csrc` or __graft_entry__.build()) through
ctypes and wraps the handle.  There is no Python or CPU rendering path: if the library is missing
or no GPU is present, the calls fail.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# ENIGMA_LIB: another build of the library (experiment builds made by tools/build_exp.sh); the product is the one beside this file
LIB_PATH = os.environ.get("ENIGMA_LIB") or os.path.join(_HERE, "libenigma_hip.so")

# every symbol include/enigma_hip.h declares
SYMBOLS = [
    "enigma_ctx_create", "enigma_ctx_destroy", "enigma_last_error", "enigma_scene_upload", "enigma_camera_set",
    "enigma_film_config", "enigma_sampler_select", "enigma_integrator_config", "enigma_direct_lighting_config", "enigma_volpath_config", "enigma_render", "enigma_film_read", "enigma_film_clear",
    "enigma_film_device_ptr", "enigma_film_resolve", "enigma_get_stats", "enigma_debug_sobol", "enigma_debug_camera_rays",
    "enigma_debug_intersect", "enigma_debug_intersect_any", "enigma_debug_primary_hits", "enigma_debug_film_footprint", "enigma_debug_intersect_noalloc", "enigma_debug_bsdf", "enigma_debug_bsdf_of", "enigma_default_metal_spectra",
    "enigma_comm_unique_id", "enigma_comm_init", "enigma_comm_info", "enigma_film_reduce", "enigma_comm_destroy",
]

_lib = None


def load_library():
    """Loads the HIP library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C raytrace_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.enigma_last_error.restype = C.c_char_p
        L.enigma_last_error.argtypes = [C.c_void_p]
        L.enigma_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.enigma_ctx_destroy.argtypes = [C.c_void_p]
        L.enigma_ctx_destroy.restype = None
        L.enigma_render.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.enigma_film_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.enigma_film_resolve.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        L.enigma_film_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        for name in ("enigma_scene_upload", "enigma_camera_set", "enigma_film_config", "enigma_integrator_config", "enigma_direct_lighting_config", "enigma_get_stats"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.enigma_film_clear.argtypes = [C.c_void_p]
        if not os.environ.get("ENIGMA_LIB") or hasattr(L, "enigma_sampler_select"):  # experiment builds of an older tree lack it
            L.enigma_sampler_select.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.enigma_comm_unique_id.argtypes = [C.c_void_p]
        L.enigma_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.enigma_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.enigma_film_reduce.argtypes = [C.c_void_p, C.c_int]
        L.enigma_comm_destroy.argtypes = [C.c_void_p]
        L.enigma_debug_sobol.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.enigma_debug_camera_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_intersect_noalloc.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_intersect_any.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.enigma_debug_primary_hits.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_bsdf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_bsdf_of.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.enigma_debug_film_footprint.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class EnigmaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("enigma error %d: %s" % (code, msg))
        self.code = code


class Renderer:
    """Thin handle wrapper: scene/camera/film/integrator descs in, film out."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.enigma_ctx_create(device, C.byref(self.h))
        if rc != 0:
            raise EnigmaError(rc, "enigma_ctx_create failed (no HIP device %d?)" % device)
        self._keep = []
        self.w = self.hgt = 0

    def _chk(self, rc):
        if rc != 0:
            raise EnigmaError(rc, self.L.enigma_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.enigma_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, scene, camera, film, integ, sampler="sobol", sample_at_pixel_center=False):
        self.upload_scene(scene)
        self.set_camera(camera)
        self.set_film(film)
        self.select_sampler(sampler, sample_at_pixel_center)
        self.set_integrator(integ)
        return self

    def select_sampler(self, name, sample_at_pixel_center=False):
        """'sobol' | 'halton' for the integrators configured from here on (RenderOptions::MakeSampler, Enigma/Api.cpp:425-434)."""
        kind = {"sobol": abi.SAMPLER_SOBOL, "halton": abi.SAMPLER_HALTON}[name]
        if kind == abi.SAMPLER_SOBOL and os.environ.get("ENIGMA_LIB") and not hasattr(self.L, "enigma_sampler_select"):
            return
        self._chk(self.L.enigma_sampler_select(self.h, kind, 1 if sample_at_pixel_center else 0))

    def upload_scene(self, scene):
        d = scene.desc()
        self._keep.append(scene)
        self._chk(self.L.enigma_scene_upload(self.h, C.byref(d)))

    def set_camera(self, camera):
        self._chk(self.L.enigma_camera_set(self.h, C.byref(camera)))

    def set_film(self, film):
        self._chk(self.L.enigma_film_config(self.h, C.byref(film)))
        b = film.cropped_pixel_bounds
        self.w, self.hgt = b[2] - b[0], b[3] - b[1]

    def set_integrator(self, integ):
        """EnigmaIntegratorDesc -> PathIntegrator, EnigmaDirectLightingDesc -> DirectLightingIntegrator, EnigmaVolPathDesc -> VolPathIntegrator."""
        if isinstance(integ, abi.EnigmaDirectLightingDesc):
            self._chk(self.L.enigma_direct_lighting_config(self.h, C.byref(integ)))
        elif isinstance(integ, abi.EnigmaVolPathDesc):
            self._chk(self.L.enigma_volpath_config(self.h, C.byref(integ)))
        else:
            self._chk(self.L.enigma_integrator_config(self.h, C.byref(integ)))

    def render(self, s0, s1):
        self._chk(self.L.enigma_render(self.h, s0, s1))

    def clear(self):
        self._chk(self.L.enigma_film_clear(self.h))

    def film(self):
        out = np.zeros((self.hgt, self.w, 4), dtype=np.float32)
        self._chk(self.L.enigma_film_read(self.h, abi.ptr(out), out.size))
        return out

    # ---- multi-GPU (RCCL inside the library) ----
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId; rank 0 makes them, every rank passes them to comm_init."""
        buf = C.create_string_buffer(abi.COMM_ID_BYTES)
        rc = load_library().enigma_comm_unique_id(buf)
        if rc != 0:
            raise EnigmaError(rc, "enigma_comm_unique_id failed (RCCL not loadable?)")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == abi.COMM_ID_BYTES
        self._chk(self.L.enigma_comm_init(self.h, rank, world, C.create_string_buffer(unique_id, abi.COMM_ID_BYTES)))

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self._chk(self.L.enigma_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def film_reduce(self, root=0):
        self._chk(self.L.enigma_film_reduce(self.h, root))

    def comm_destroy(self):
        self._chk(self.L.enigma_comm_destroy(self.h))

    def film_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.enigma_film_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def image(self, scale=1.0, film=None):
        return resolve(self.film() if film is None else film, scale)

    def stats(self):
        s = abi.EnigmaStats()
        self._chk(self.L.enigma_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in abi.EnigmaStats._fields_}

    # ---- debug taps ----
    def sobol(self, pixels, sample_nums, n_dims):
        pixels = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        sn = np.ascontiguousarray(sample_nums, dtype=np.int64)
        n = pixels.shape[0]
        idx = np.zeros(n, dtype=np.uint64)
        dims = np.zeros((n, n_dims), dtype=np.float32)
        self._chk(self.L.enigma_debug_sobol(self.h, n, abi.ptr(pixels), abi.ptr(sn), n_dims, abi.ptr(idx), abi.ptr(dims)))
        return idx, dims

    def camera_rays(self, pixels, sample_nums):
        pixels = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        sn = np.ascontiguousarray(sample_nums, dtype=np.int64)
        n = pixels.shape[0]
        rays = np.zeros((n, 6), dtype=np.float32)
        pf = np.zeros((n, 2), dtype=np.float32)
        self._chk(self.L.enigma_debug_camera_rays(self.h, n, abi.ptr(pixels), abi.ptr(sn), abi.ptr(rays), abi.ptr(pf)))
        return rays, pf

    def intersect(self, rays7, noalloc=False):
        """Scene::intersect for rays (o, d, tMax); noalloc=True goes through the entry point that stages in the
        library's fixed device array instead of allocating."""
        rays7 = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        n = rays7.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        pid = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.float32)
        bary = np.zeros((n, 3), dtype=np.float32)
        fn = self.L.enigma_debug_intersect_noalloc if noalloc else self.L.enigma_debug_intersect
        self._chk(fn(self.h, n, abi.ptr(rays7), abi.ptr(tri), abi.ptr(pid), abi.ptr(t), abi.ptr(bary)))
        return dict(tri=tri, prim_id=pid, t=t, bary=bary)

    def primary_hits(self, sample_num, region_wh):
        """The hit records the production closest-hit kernel leaves for the camera rays of one sample number, as (h, w)
        arrays over the render region (region_wh = its width and height)."""
        w, hgt = region_wh
        n = w * hgt
        tri = np.zeros(n, dtype=np.int32)
        pid = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.float32)
        bary = np.zeros((n, 3), dtype=np.float32)
        self._chk(self.L.enigma_debug_primary_hits(self.h, sample_num, n, abi.ptr(tri), abi.ptr(pid), abi.ptr(t), abi.ptr(bary)))
        return dict(tri=tri.reshape(hgt, w), prim_id=pid.reshape(hgt, w), t=t.reshape(hgt, w), bary=bary.reshape(hgt, w, 3))

    def intersect_any(self, rays7):
        rays7 = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(rays7.shape[0], dtype=np.int32)
        self._chk(self.L.enigma_debug_intersect_any(self.h, rays7.shape[0], abi.ptr(rays7), abi.ptr(out)))
        return out

    def bsdf_eval(self, material, wo, wi, u):
        """f, Pdf and sample_f of the material's BSDF at a +z surface for (wo, wi, u) triples: an (n, 12) array."""
        wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((wo.shape[0], 12), dtype=np.float32)
        self._chk(self.L.enigma_debug_bsdf(self.h, C.byref(material), wo.shape[0], abi.ptr(wo), abi.ptr(wi), abi.ptr(u), abi.ptr(out)))
        return out

    def bsdf_eval_of(self, materials, which, wo, wi, u, multiple_lobes=True):
        """bsdf_eval for entry `which` of a list of materials (a mix names other entries), with allowMultipleLobes as given."""
        wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((wo.shape[0], 12), dtype=np.float32)
        table = (abi.EnigmaMaterial * len(materials))(*materials)
        self._chk(self.L.enigma_debug_bsdf_of(self.h, table, len(materials), which, int(bool(multiple_lobes)), wo.shape[0], abi.ptr(wo), abi.ptr(wi),
                                              abi.ptr(u), abi.ptr(out)))
        return out

    def film_footprint(self, pfilm, max_per=64):
        pfilm = np.ascontiguousarray(pfilm, dtype=np.float32).reshape(-1, 2)
        n = pfilm.shape[0]
        pix = np.zeros((n, max_per, 2), dtype=np.int32)
        w = np.zeros((n, max_per), dtype=np.float32)
        cnt = np.zeros(n, dtype=np.uint32)
        self._chk(self.L.enigma_debug_film_footprint(self.h, n, abi.ptr(pfilm), max_per, abi.ptr(pix), abi.ptr(w), abi.ptr(cnt)))
        return pix, w, cnt


def default_metal_spectra():
    """(eta rgb, k rgb) of a metal that names neither: the reference's copper (host code inside the library)."""
    L = load_library()
    eta, k = (C.c_float * 3)(), (C.c_float * 3)()
    rc = L.enigma_default_metal_spectra(eta, k)
    if rc != 0:
        raise EnigmaError(rc, "enigma_default_metal_spectra")
    return np.array(eta[:], dtype=np.float32), np.array(k[:], dtype=np.float32)


def resolve(film_rgbw, scale=1.0):
    """Film::WriteImage normalisation of an rgbw film (host code inside the library)."""
    L = load_library()
    f = np.ascontiguousarray(film_rgbw, dtype=np.float32)
    out = np.zeros(f.shape[:-1] + (3,), dtype=np.float32)
    rc = L.enigma_film_resolve(abi.ptr(f), f.size // 4, scale, abi.ptr(out))
    if rc != 0:
        raise EnigmaError(rc, "enigma_film_resolve")
    return out
