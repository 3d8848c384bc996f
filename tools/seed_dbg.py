"""One fuzz seed, one sample number at a time, GPU against oracle: ray counts and the three worst pixels per sample (needs a GPU).
    python tools/seed_dbg.py <seed>"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import orc, fuzz_scenes
import raytrace_amd as ra
seed = int(sys.argv[1])
d = {}
sc, cam, film, integ, label = fuzz_scenes.random_case(seed, details=d)
print(label)
for s in range(d["spp"]):
    r = ra.Renderer(0).configure(sc, cam, film, integ); r.render(s, s + 1)
    o = orc.Oracle(sc, cam, film, integ); o.render(s, s + 1, threads=8)
    a, b = r.image(), o.image()
    e = np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), 1e-3)
    idx = np.argsort(e.reshape(-1))[::-1][:3]
    print("sample", s, "rays", {k: (r.stats()[k], o.stats()[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")})
    for i in idx:
        y, x, c = np.unravel_index(i, e.shape)
        print("  ", y, x, c, "gpu", a[y, x], "oracle", b[y, x], "err", e[y, x, c])
    r.close(); o.close()
