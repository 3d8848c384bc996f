"""BASELINE.json configs[0] (Cornell 512 x 512, 16 spp, the reference's own CPU-runnable case) and configs[3] at its stated
size (ManyLights 512 x 512, 16 spp): the library on cuda:0 and the CPU oracle on every usable core, one JSON line each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/..  
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # the oracle is test infrastructure: this script lives with the tests
import numpy as np
import raytrace_amd as ra
from raytrace_amd import scenes
import orc

threads = len(os.sched_getaffinity(0))
for name, scene in (("cornell_microfacet_512x512x16", scenes.cornell_box("microfacet")), ("manylights_512x512x16", scenes.many_lights())):
    cam, film = scenes.cornell_camera(512, 512), scenes.film_desc(512, 512)
    integ = scenes.integrator_desc(film, 16)
    r = ra.Renderer(0).configure(scene, cam, film, integ)
    r.render(0, 16)                      # warm-up
    best = None
    for _ in range(5):
        r.clear_film() if hasattr(r, "clear_film") else None
        s0 = r.stats(); t0 = time.perf_counter(); r.render(0, 16); dt = time.perf_counter() - t0; s1 = r.stats()
        rays = sum(s1[k] - s0[k] for k in ("rays_closest", "rays_shadow", "rays_mis"))
        if best is None or dt < best[0]:
            best = (dt, rays)
    gpu_img = None
    r2 = ra.Renderer(0).configure(scene, cam, film, integ); r2.render(0, 16); gpu_img = r2.image(); r2.close(); r.close()
    o = orc.Oracle(scene, cam, film, integ)
    t0 = time.perf_counter(); o.render(0, 16, threads=threads); cdt = time.perf_counter() - t0
    st = o.stats(); crays = st["rays_closest"] + st["rays_shadow"] + st["rays_mis"]
    img = o.image(); o.close()
    mre = float(np.mean(np.abs(gpu_img - img) / np.maximum(img, 1e-3)))
    n = 512 * 512 * 16
    print(json.dumps({"config": name, "gpu_ms": best[0] * 1e3, "gpu_mrays_s": best[1] / best[0] / 1e6, "gpu_msamples_s": n / best[0] / 1e6,
                      "cpu_s": cdt, "cpu_threads": threads, "cpu_mrays_s": crays / cdt / 1e6, "cpu_msamples_s": n / cdt / 1e6, "mre_gpu_vs_cpu": mre}))
