#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so, strict IEEE build).

The reference holds no golden vectors of its own (no tests, no images), so these fixtures are
outputs of the restated reference algorithm on fixed inputs; the inputs are stored beside the
outputs.  They pin the oracle against regressions and give the GPU tests data that does not need
the oracle at run time.  Floats are stored as uint32 bit patterns where bit-exactness is claimed.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, orc
from raytrace_amd import abi, scenes

G = os.path.join(ROOT, "tests", "golden")
u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

def sobol_and_camera():
    out = {}
    for name, (xr, yr, spp) in {"m9": (512, 512, 16), "m11": (1920, 1080, 1024)}.items():
        sc, cam, film, integ = common.cornell_setup(xres=xr, yres=yr, spp=spp)
        o = orc.Oracle(sc, cam, film, integ)
        pix = [(0, 0), (1, 0), (xr - 1, yr - 1), (777 % xr, 333 % yr), (xr // 2, yr // 2), (xr - 1, 0), (0, yr - 1)]
        sns = [0, 1, 2, 15, spp - 1]
        P = np.array([p for p in pix for _ in sns], dtype=np.int32)
        S = np.array([s for _ in pix for s in sns], dtype=np.int64)
        idx, dims = o.sobol(P, S, 8)
        rays, pf = o.camera_rays(P, S)
        out[name + "_pixels"] = P; out[name + "_samples"] = S; out[name + "_index"] = idx
        out[name + "_dims_bits"] = u32(dims); out[name + "_rays_bits"] = u32(rays); out[name + "_pfilm_bits"] = u32(pf)
        out[name + "_camera"] = np.concatenate([np.array(cam.raster_to_camera[:]), np.array(cam.camera_to_world[:])]).astype(np.float32)
        o.close()
    np.savez_compressed(os.path.join(G, "sobol_camera.npz"), **out)

def primary_hits():
    out = {}
    for variant in ("diffuse",):
        sc, cam, film, integ = common.cornell_setup(res=64, spp=16, variant=variant)
        o = orc.Oracle(sc, cam, film, integ)
        ys, xs = np.mgrid[0:64, 0:64]
        P = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
        for sn in (0, 7):
            S = np.full(P.shape[0], sn, dtype=np.int64)
            rays, _ = o.camera_rays(P, S)
            h = o.intersect(common.rays_from_camera(rays))
            out["%s_s%d_rays_bits" % (variant, sn)] = u32(rays)
            out["%s_s%d_prim_id" % (variant, sn)] = h["prim_id"]
            out["%s_s%d_t_bits" % (variant, sn)] = u32(h["t"])
            out["%s_s%d_bary_bits" % (variant, sn)] = u32(h["bary"])
            out["%s_s%d_ambiguous" % (variant, sn)] = h["ambiguous"]
        o.close()
    np.savez_compressed(os.path.join(G, "primary_hits.npz"), **out)

def film_footprints():
    out = {}
    rng = np.random.RandomState(7)
    pf = np.concatenate([rng.uniform(-1, 66, size=(24, 2)), np.array([[0, 0], [10, 10], [10.5, 10.5], [63.999, 63.999], [64, 64], [31.0, 17.25], [0.5, 63.5], [5.0, 5.5]])]).astype(np.float32)
    for kind in ("box", "triangle", "gaussian", "mitchell", "sinc"):
        sc, cam, film, integ = common.cornell_setup(res=64, spp=16, filter_kind=kind)
        o = orc.Oracle(sc, cam, film, integ)
        pix, w, cnt = o.film_footprint(pf, 32)
        out[kind + "_pix"] = pix; out[kind + "_w_bits"] = u32(w); out[kind + "_cnt"] = cnt
        out[kind + "_table_bits"] = u32(np.array(film.filter_table[:], dtype=np.float32))
        o.close()
    out["pfilm_bits"] = u32(pf)
    np.savez_compressed(os.path.join(G, "film_footprints.npz"), **out)

def bsdf_points():
    out = {}
    rng = np.random.RandomState(11)
    def dirs(n):
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True); return v.astype(np.float32)
    wo, wi, u = dirs(16), dirs(16), rng.uniform(0, 1, size=(16, 2)).astype(np.float32)
    mats = {"matte": abi.matte((0.6, 0.5, 0.4)), "metal": abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, 0.1),
            "glass_rough": abi.glass(1.5, 0.05, 0.08), "glass_smooth": abi.glass(1.5, 0, 0), "plastic": abi.plastic((0.3, 0.3, 0.6), (0.4, 0.4, 0.4), 0.05)}
    for k, m in mats.items():
        out[k] = orc.bsdf_eval(m, wo, wi, u)
    out["wo"], out["wi"], out["u"] = wo, wi, u
    np.savez_compressed(os.path.join(G, "bsdf_points.npz"), **out)

def images():
    out = {}
    for variant in ("diffuse", "microfacet", "specular"):
        sc, cam, film, integ = common.cornell_setup(res=64, spp=64, variant=variant)
        o = orc.Oracle(sc, cam, film, integ)
        o.render(0, 64, threads=8, moments=True)
        out[variant + "_rgb"] = o.image()
        out[variant + "_moments"] = o.moments().astype(np.float32)
        s = o.stats()
        out[variant + "_rays"] = np.array([s["samples"], s["rays_closest"], s["rays_shadow"], s["rays_mis"]], dtype=np.int64)
        o.close()
    np.savez_compressed(os.path.join(G, "cornell64_64spp.npz"), **out)

def direct_lighting_images():
    """DirectLightingIntegrator: strategy all (two lights, 2 + 1 samples) and strategy one through smooth glass."""
    from raytrace_amd import abi, scenes
    out = {}
    sc, cam, film, _ = common.cornell_setup(res=64, spp=16, variant="microfacet")
    sc.lights[0].n_samples = 2
    o = orc.Oracle(sc, cam, film, scenes.direct_lighting_desc(film, 16, max_depth=5, strategy=abi.DL_SAMPLE_ALL))
    o.render(0, 16, threads=8)
    s = o.stats()
    out["all_rgb"] = o.image()
    out["all_rays"] = np.array([s["samples"], s["rays_closest"], s["rays_shadow"], s["rays_mis"]], dtype=np.int64)
    o.close()
    sc, cam, film, _ = common.cornell_setup(res=64, spp=16, variant="specular")
    o = orc.Oracle(sc, cam, film, scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ONE))
    o.render(0, 16, threads=8)
    s = o.stats()
    out["one_rgb"] = o.image()
    out["one_rays"] = np.array([s["samples"], s["rays_closest"], s["rays_shadow"], s["rays_mis"]], dtype=np.int64)
    o.close()
    np.savez_compressed(os.path.join(G, "direct_lighting64_16spp.npz"), **out)


def delta_light_images():
    """PathIntegrator over the Cornell box with a spot and a distant light added (power / uniform light sampling)."""
    out = {}
    for variant, strategy in (("diffuse", abi.LIGHTS_POWER), ("microfacet", abi.LIGHTS_UNIFORM)):
        sc = scenes.cornell_box(variant, delta_lights=True)
        cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
        o = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, 16, light_strategy=strategy))
        o.render(0, 16, threads=8)
        s = o.stats()
        out[variant + "_rgb"] = o.image()
        out[variant + "_rays"] = np.array([s["samples"], s["rays_closest"], s["rays_shadow"], s["rays_mis"]], dtype=np.int64)
        o.close()
    np.savez_compressed(os.path.join(G, "delta_lights64_16spp.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "direct":
        direct_lighting_images()
    elif len(sys.argv) > 1 and sys.argv[1] == "delta":
        delta_light_images()
    elif len(sys.argv) > 1 and sys.argv[1] == "film":
        film_footprints()
    else:
        sobol_and_camera(); primary_hits(); film_footprints(); bsdf_points(); images(); direct_lighting_images(); delta_light_images()
    for f in sorted(os.listdir(G)):
        print(f, os.path.getsize(os.path.join(G, f)))
