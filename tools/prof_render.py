"""Profiling driver: a few 16-spp steps of the bench workload (no torch, no oracle) for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_amd as ra
from raytrace_amd import scenes, abi

variant = sys.argv[1] if len(sys.argv) > 1 else "microfacet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
xres, yres = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
if variant == "bigmesh":
    scene, camera = scenes.big_mesh(), scenes.big_mesh_camera(xres, yres)
elif variant == "bigmesh_fog":
    scene, camera = scenes.big_mesh_fog(), scenes.big_mesh_camera(xres, yres)
elif variant == "manylights":
    scene, camera = scenes.many_lights(), scenes.cornell_camera(xres, yres)
else:
    scene, camera = (scenes.cornell_fog("microfacet") if variant == "fog" else scenes.cornell_box(variant)), scenes.cornell_camera(xres, yres)
film = scenes.film_desc(xres, yres, "box")
integ = scenes.volpath_desc(film, 512) if variant in ("bigmesh_fog", "fog") else scenes.integrator_desc(film, 1024)
r = ra.Renderer(0).configure(scene, camera, film, integ)
for i in range(steps):
    r.render(i * 16, i * 16 + 16)
st = r.stats()
rays = st["rays_closest"] + st["rays_shadow"] + st["rays_mis"]
print("steps", steps, "last ms_render", st["ms_render"], "ms_intersect", st["ms_intersect"], "rays", rays, "Mrays/s(last step)", rays / steps / st["ms_render"] / 1e3,
      "bvh nodes", st["bvh_nodes"], "depth", st["bvh_depth"], "build ms", st["ms_bvh_build"])
r.close()
