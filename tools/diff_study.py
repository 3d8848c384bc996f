"""How do GPU and oracle images differ on BigMesh?  Counts differing pixels and the size of the differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytrace_amd as ra
from raytrace_amd import scenes, abi
import orc
xr, yr, spp = 256, 144, 4
md = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene, camera = scenes.big_mesh(), scenes.big_mesh_camera(xr, yr)
film = scenes.film_desc(xr, yr, "box")
integ = scenes.integrator_desc(film, 16, max_depth=md)
r = ra.Renderer(0).configure(scene, camera, film, integ)
o = orc.Oracle(scene, camera, film, integ, fast=False)
r.render(0, spp); o.render(0, spp, threads=16)
g, c = r.image(), o.image()
d = np.abs(g - c).max(axis=2)
rel = d / np.maximum(c.max(axis=2), 1e-3)
print("max_depth", md, "pixels", d.size, "differing >1e-4 rel:", int((rel > 1e-4).sum()), ">1e-2:", int((rel > 1e-2).sum()), "mre", float(np.mean(np.abs(g - c) / np.maximum(c, 1e-3))))
sg, so = r.stats(), o.stats()
print("rays gpu", sg["rays_closest"], sg["rays_shadow"], sg["rays_mis"], "oracle", so["rays_closest"], so["rays_shadow"], so["rays_mis"])
# primary hits: every pixel centre sample 0
ys, xs = np.mgrid[0:yr, 0:xr]
P = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
rays, _ = o.camera_rays(P, np.zeros(P.shape[0], dtype=np.int64))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
r7 = common.rays_from_camera(rays)
ho, hg = o.intersect(r7), r.intersect(r7)
neq = ho["prim_id"] != hg["prim_id"]
print("primary hits differing:", int(neq.sum()), "of", len(neq), "ambiguous flagged:", int((ho["ambiguous"] != 0).sum()), "differing & not ambiguous:", int((neq & (ho["ambiguous"] == 0)).sum()))
