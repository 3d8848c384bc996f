"""Experiment check: renders a few samples of the bench scenes with the library ENIGMA_LIB names and saves films, primary
hits and ray counts; `python tools/exp_film.py --compare a.npz b.npz` compares two such files (ray counts and hits must be
equal, films equal up to the order of the few atomic adds on pixel edges)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        if k.endswith("_film"):
            d = np.abs(a[k] - b[k]) / np.maximum(np.abs(a[k]), 1e-3)
            good = d.max() < 1e-5
            print(k, "max rel diff %.3g" % d.max(), "ok" if good else "DIFFERENT")
        else:
            good = np.array_equal(a[k], b[k])
            print(k, "equal" if good else "DIFFERENT (%d of %d)" % ((a[k] != b[k]).sum(), a[k].size))
        ok = ok and good
    sys.exit(0 if ok else 1)

import raytrace_amd as ra
from raytrace_amd import scenes
out = {}
for name, scene, cam, spp in (("cornell", scenes.cornell_box("microfacet"), scenes.cornell_camera(640, 360), 8),
                              ("bigmesh", scenes.big_mesh(), scenes.big_mesh_camera(640, 360), 4)):
    film = scenes.film_desc(640, 360)
    integ = scenes.integrator_desc(film, 64)
    r = ra.Renderer(0).configure(scene, cam, film, integ)
    hits = r.primary_hits(3, (640, 360))
    r.render(0, spp)
    st = r.stats()
    out[name + "_film"] = r.film()
    out[name + "_rays"] = np.array([st["rays_closest"], st["rays_shadow"], st["rays_mis"], st["stack_overflows"]], dtype=np.int64)
    out[name + "_prim"] = hits["prim_id"]
    out[name + "_t"] = hits["t"].view(np.uint32)
    out[name + "_bary"] = hits["bary"].view(np.uint32)
    # secondary rays through the intersection taps (any-hit and closest-hit), from the primary hit points in random directions
    rng = np.random.RandomState(7)
    rays, _ = r.camera_rays(np.stack(np.meshgrid(np.arange(0, 640, 4), np.arange(0, 360, 4)), -1).reshape(-1, 2), np.zeros(160 * 90, dtype=np.int64))
    h0 = r.intersect(np.concatenate([rays, np.full((rays.shape[0], 1), np.inf, np.float32)], axis=1))
    org = rays[:, :3] + rays[:, 3:6] * (np.where(np.isfinite(h0["t"]), h0["t"], 1.0)[:, None] * 0.999)
    sec = np.concatenate([org, rng.normal(size=org.shape), rng.uniform(1, 600, (org.shape[0], 1))], axis=1).astype(np.float32)
    out[name + "_any"] = r.intersect_any(sec)
    r.close()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])
