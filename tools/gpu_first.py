"""Dev script: first contact with the GPU -- taps + a small render compared with the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytrace_amd as ra
from raytrace_amd import scenes, abi
import orc, common

res, spp = 128, 16
for variant in ("diffuse", "microfacet", "specular"):
    sc, cam, film, integ = common.cornell_setup(res, spp, variant)
    o = orc.Oracle(sc, cam, film, integ)
    r = ra.Renderer().configure(sc, cam, film, integ)
    rng = np.random.RandomState(1)
    px = rng.randint(0, res, size=(2000, 2)); sn = rng.randint(0, spp, size=2000)
    io, do = o.sobol(px, sn, 20); ig, dg = r.sobol(px, sn, 20)
    print(variant, "sobol idx eq", np.array_equal(io, ig), "dims eq", np.array_equal(do.view(np.uint32), dg.view(np.uint32)))
    ro, po = o.camera_rays(px, sn); rg, pg = r.camera_rays(px, sn)
    print(" camera eq", np.array_equal(ro.view(np.uint32), rg.view(np.uint32)), np.array_equal(po.view(np.uint32), pg.view(np.uint32)))
    ho = o.intersect(common.rays_from_camera(ro)); hg = r.intersect(common.rays_from_camera(ro))
    print(" hits eq tri", np.mean(ho["tri"] == hg["tri"]), "t", np.mean(ho["t"].view(np.uint32) == hg["t"].view(np.uint32)), "amb", int((ho["ambiguous"] > 0).sum()))
    t0 = time.time(); r.render(0, spp); t1 = time.time()
    img_g = r.image()
    o.render(0, spp, threads=0); img_o = o.image()
    st = r.stats(); so = o.stats()
    print(" render s", t1 - t0, "gpu stats", {k: st[k] for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_mis", "ms_render", "ms_intersect")})
    print(" oracle stats", {k: so[k] for k in ("samples", "rays_closest", "rays_shadow", "rays_mis")})
    print(" mean img", img_g.mean(axis=(0, 1)), img_o.mean(axis=(0, 1)), "MRE", common.mean_relative_error(img_g, img_o), "max abs", np.abs(img_g - img_o).max())
    r.close(); o.close()
