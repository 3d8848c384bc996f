"""ctypes binding of the CPU oracle (oracle/liboracle*.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

from raytrace_amd import abi, sobol_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_libs = {}
_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = sobol_tables.generate()
    return _tables


def lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _libs:
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), name])
        L = C.CDLL(path)
        L.oracle_last_error.restype = C.c_char_p
        for fn in ("oracle_gamma", "oracle_next_float_up", "oracle_next_float_down", "oracle_fr_dielectric", "oracle_tr_D"):
            getattr(L, fn).restype = C.c_float
        L.oracle_gamma.argtypes = [C.c_int]
        L.oracle_next_float_up.argtypes = [C.c_float]
        L.oracle_next_float_down.argtypes = [C.c_float]
        L.oracle_fr_dielectric.argtypes = [C.c_float] * 3
        L.oracle_tr_D.argtypes = [C.c_float] * 5
        L.oracle_distribution1d.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_rgb_from_sampled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_default_metal.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_spectral_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_film_write_image.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
        L.oracle_scene_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        m32, vdc, vdci = tables()
        L.oracle_set_sobol_tables(abi.ptr(m32), abi.ptr(vdc), abi.ptr(vdci))
        _libs[name] = L
    return _libs[name]


def halton_tables(n_words=10000, n_perm=100000):
    """(first words of the engine behind the oracle's RNG, head of its digit-permutation table, primes, prime sums, table length)"""
    words, perm = np.zeros(n_words, dtype=np.uint32), np.zeros(n_perm, dtype=np.uint16)
    primes, sums = np.zeros(1000, dtype=np.int32), np.zeros(1000, dtype=np.int32)
    L = lib()
    L.oracle_debug_halton_tables.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    total = L.oracle_debug_halton_tables(n_words, abi.ptr(words), n_perm, abi.ptr(perm), abi.ptr(primes), abi.ptr(sums))
    return words, perm, primes, sums, total


def rgb_from_sampled(lam, v):
    """RGBSpectrum::FromSampled of the restatement."""
    lam = np.ascontiguousarray(lam, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_rgb_from_sampled(abi.ptr(lam), abi.ptr(v), C.c_int(lam.shape[0]), abi.ptr(out))
    return out


def default_metal():
    eta, k = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
    lib().oracle_default_metal(abi.ptr(eta), abi.ptr(k))
    return eta, k


def spectral_tables():
    cie, cu = np.zeros((4, 471), dtype=np.float32), np.zeros((3, 56), dtype=np.float32)
    lib().oracle_spectral_tables(abi.ptr(cie), abi.ptr(cu))
    return cie, cu


class Oracle:
    def __init__(self, scene, camera, film, integ, fast=False, split_method="hlbvh", max_node_prims=128, sampler="sobol", sample_at_pixel_center=False):
        self.L = lib(fast)
        self.h = C.c_void_p()
        self.L.oracle_create(C.byref(self.h))
        self.scene, self.camera, self.film, self.integ = scene, camera, film, integ
        d = scene.desc()
        self._chk(self.L.oracle_scene_upload(self.h, C.byref(d), 1 if split_method == "hlbvh" else 0, max_node_prims))
        self._chk(self.L.oracle_camera_set(self.h, C.byref(camera)))
        self._chk(self.L.oracle_film_config(self.h, C.byref(film)))
        self._chk(self.L.oracle_sampler_select(self.h, {"sobol": 0, "halton": 1}[sampler], 1 if sample_at_pixel_center else 0))
        self.set_integrator(integ)
        b = film.cropped_pixel_bounds
        self.w, self.hgt = b[2] - b[0], b[3] - b[1]

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.oracle_last_error(self.h).decode())

    def set_integrator(self, integ):
        if isinstance(integ, abi.EnigmaDirectLightingDesc):
            self._chk(self.L.oracle_direct_lighting_config(self.h, C.byref(integ)))
        elif isinstance(integ, abi.EnigmaVolPathDesc):
            self._chk(self.L.oracle_volpath_config(self.h, C.byref(integ)))
        else:
            self._chk(self.L.oracle_integrator_config(self.h, C.byref(integ)))
        self.integ = integ

    def read_past_arrays(self):
        """True if a render since the last call read past the integrator's sample arrays (undefined in the reference)."""
        return bool(self.L.oracle_read_past_arrays(self.h))

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def render(self, s0, s1, threads=0, moments=False):
        self._chk(self.L.oracle_render(self.h, s0, s1, threads, 1 if moments else 0))

    def image(self, scale=1.0):
        out = np.zeros((self.hgt, self.w, 3), dtype=np.float32)
        self.L.oracle_film_write_image(self.h, scale, abi.ptr(out))
        return out

    def film_xyzw(self):
        out = np.zeros((self.hgt, self.w, 4), dtype=np.float32)
        self.L.oracle_film_read_xyzw(self.h, abi.ptr(out))
        return out

    def moments(self):
        out = np.zeros((self.hgt, self.w, 7), dtype=np.float64)
        self.L.oracle_read_moments(self.h, abi.ptr(out))
        return out

    def clear(self):
        self.L.oracle_film_clear(self.h)

    def stats(self):
        out = (C.c_uint64 * 8)()
        ms = C.c_double()
        self.L.oracle_get_stats(self.h, out, C.byref(ms))
        keys = ["samples", "rays_closest", "rays_shadow", "rays_mis", "nodes", "tris", "bvh_nodes", "degenerate_partitions"]
        d = dict(zip(keys, [int(v) for v in out]))
        d["ms_render"] = ms.value
        k = (C.c_uint64 * 6)()
        self.L.oracle_get_kind_stats(self.h, k)
        for i, name in enumerate(("closest", "shadow", "mis")):
            d["nodes_" + name] = int(k[i])
            d["tris_" + name] = int(k[3 + i])
        return d

    def reset_stats(self):
        self.L.oracle_reset_stats(self.h)

    def sobol(self, pixels, sample_nums, n_dims):
        pixels = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        sn = np.ascontiguousarray(sample_nums, dtype=np.int64)
        n = pixels.shape[0]
        idx = np.zeros(n, dtype=np.uint64)
        dims = np.zeros((n, n_dims), dtype=np.float32)
        self.L.oracle_debug_sobol(self.h, n, abi.ptr(pixels), abi.ptr(sn), n_dims, abi.ptr(idx), abi.ptr(dims))
        return idx, dims

    def camera_rays(self, pixels, sample_nums):
        pixels = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        sn = np.ascontiguousarray(sample_nums, dtype=np.int64)
        n = pixels.shape[0]
        rays = np.zeros((n, 6), dtype=np.float32)
        pf = np.zeros((n, 2), dtype=np.float32)
        self.L.oracle_debug_camera_rays(self.h, n, abi.ptr(pixels), abi.ptr(sn), abi.ptr(rays), abi.ptr(pf))
        return rays, pf

    def intersect(self, rays7, ambiguous=True):
        rays7 = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        n = rays7.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        pid = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.float32)
        bary = np.zeros((n, 3), dtype=np.float32)
        nodes = np.zeros(n, dtype=np.uint32)
        tris = np.zeros(n, dtype=np.uint32)
        amb = np.zeros(n, dtype=np.uint32) if ambiguous else None
        self.L.oracle_debug_intersect(self.h, n, abi.ptr(rays7), abi.ptr(tri), abi.ptr(pid), abi.ptr(t), abi.ptr(bary),
                                      abi.ptr(nodes), abi.ptr(tris), abi.ptr(amb))
        return dict(tri=tri, prim_id=pid, t=t, bary=bary, nodes=nodes, tris=tris, ambiguous=amb)

    def intersect_any(self, rays7):
        rays7 = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(rays7.shape[0], dtype=np.int32)
        self.L.oracle_debug_intersect_any(self.h, rays7.shape[0], abi.ptr(rays7), abi.ptr(out))
        return out

    def film_footprint(self, pfilm, max_per=64):
        pfilm = np.ascontiguousarray(pfilm, dtype=np.float32).reshape(-1, 2)
        n = pfilm.shape[0]
        pix = np.zeros((n, max_per, 2), dtype=np.int32)
        w = np.zeros((n, max_per), dtype=np.float32)
        cnt = np.zeros(n, dtype=np.uint32)
        self.L.oracle_debug_film_footprint(self.h, n, abi.ptr(pfilm), max_per, abi.ptr(pix), abi.ptr(w), abi.ptr(cnt))
        return pix, w, cnt


def bsdf_eval(material, wo, wi, u, fast=False):
    L = lib(fast)
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((wo.shape[0], 12), dtype=np.float32)
    L.oracle_debug_bsdf(C.byref(material), wo.shape[0], abi.ptr(wo), abi.ptr(wi), abi.ptr(u), abi.ptr(out))
    return out


def bsdf_eval_of(materials, which, wo, wi, u, multiple_lobes=True, fast=False):
    """bsdf_eval for entry `which` of a list of materials (a mix names other entries), with allowMultipleLobes as given."""
    L = lib(fast)
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((wo.shape[0], 12), dtype=np.float32)
    table = (abi.EnigmaMaterial * len(materials))(*materials)
    rc = L.oracle_debug_bsdf_of(table, len(materials), which, int(bool(multiple_lobes)), wo.shape[0], abi.ptr(wo), abi.ptr(wi), abi.ptr(u), abi.ptr(out))
    if rc != 0:
        raise RuntimeError("oracle_debug_bsdf_of refused the material table")
    return out
