"""enigma_debug_intersect_noalloc against enigma_debug_intersect (GPU, through the C ABI).

The entry point stages rays and results in a fixed device array of the library instead of allocating
per call.  Two calls in a row share that array; the second is smaller, so stale words of the first
call sit behind its results and must not leak into them.  Bit-exact: the two entry points run the
same kernel on the same rays.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import common
import raytrace_amd as ra
from raytrace_amd import scenes

pytestmark = pytest.mark.gpu

MAX_RAYS = 131072  # ENIGMA_NOALLOC_MAX_RAYS


def _rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    tmax = np.where(rng.rand(n) < 0.25, rng.uniform(50.0, 400.0, n), np.inf).astype(np.float32)
    return np.concatenate([o, d, tmax[:, None]], axis=1).astype(np.float32)


def _same(a, b):
    for k in ("tri", "prim_id"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("t", "bary"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def _free_device_bytes(r):
    """hipMemGetInfo through the HIP runtime the library itself is linked to."""
    import ctypes as C
    free, total = C.c_size_t(), C.c_size_t()
    assert r.L.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_no_alloc_intersect_matches_the_allocating_entry_point():
    rng = np.random.RandomState(77)
    sc, cam, film, integ = common.cornell_setup(res=64, spp=16, variant="microfacet")
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    first = _rays(rng, 50000, 0.0, 556.0)
    second = _rays(rng, 1237, 0.0, 556.0)   # smaller: the first call's results lie behind it in the shared array
    want_first, want_second = r.intersect(first), r.intersect(second)
    assert (want_first["tri"] >= 0).mean() > 0.5
    _same(r.intersect(first, noalloc=True), want_first)
    before = _free_device_bytes(r)
    _same(r.intersect(second, noalloc=True), want_second)
    assert _free_device_bytes(r) == before      # nothing was allocated (or left allocated) on the device by the call
    # and a larger one again, up to the limit; one ray more is refused
    full = _rays(rng, MAX_RAYS, 0.0, 556.0)
    _same(r.intersect(full, noalloc=True), r.intersect(full))
    with pytest.raises(ra.EnigmaError) as e:
        r.intersect(_rays(rng, MAX_RAYS + 1, 0.0, 556.0), noalloc=True)
    assert e.value.code == -1
    # a second handle on the same GPU uses the same array, one call after the other
    sc2 = scenes.many_lights()
    r2 = ra.Renderer(0).configure(sc2, scenes.cornell_camera(64, 64), film, integ)
    third = _rays(rng, 4096, 0.0, 556.0)
    _same(r2.intersect(third, noalloc=True), r2.intersect(third))
    _same(r.intersect(second, noalloc=True), want_second)
    r.close(); r2.close()
