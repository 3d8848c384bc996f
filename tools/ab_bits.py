"""Bit-level A/B of two builds of the library (no oracle): BSDF tap outputs of the fuzzer's materials and directions, and the
films / ray counts of its random scenes.  `python tools/ab_bits.py dump <out.npz> <first seed> <count>` with ENIGMA_LIB naming the
build; `python tools/ab_bits.py compare a.npz b.npz`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        x, y = a[k], b[k]
        same = x.shape == y.shape and np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
        if not same:
            bad += 1
            d = np.abs(x.astype(np.float64) - y.astype(np.float64)) if x.shape == y.shape else None
            print(k, "DIFFERENT", "" if d is None else "max abs %.3g, %d of %d words" % (np.nanmax(d), int((x.view(np.uint32) != y.view(np.uint32)).sum()) if x.dtype == np.float32 else int((x != y).sum()), x.size))
    print("%d of %d arrays differ" % (bad, len(a.files)))
    sys.exit(1 if bad else 0)

import raytrace_amd as ra
import fuzz_scenes as fz
out = {}
first, count = int(sys.argv[3]), int(sys.argv[4])
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    m = fz.random_material(rng)
    wo, wi, u = fz.adversarial_directions(rng, 2000)
    r = ra.Renderer(0)
    out["bsdf_%d" % seed] = r.bsdf_eval(m, wo, wi, u)
    r.close()
    try:
        scene, cam, film, integ, label = fz.random_case(seed)
        r = ra.Renderer(0).configure(scene, cam, film, integ)
        r.render(0, 4)
        st = r.stats()
        out["film_%d" % seed] = r.film()
        out["rays_%d" % seed] = np.array([st["rays_closest"], st["rays_shadow"], st["rays_mis"]], dtype=np.int64)
        r.close()
    except ra.EnigmaError:
        pass
np.savez(sys.argv[2], **out)
print("saved", sys.argv[2], len(out), "arrays")
