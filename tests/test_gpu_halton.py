"""GPU parity for the HaltonSampler (Enigma/Sampler.cpp:305-375; the reference's default sampler, Enigma/Api.h:82) through the
C ABI; the CPU oracle is the checker.  Sample indices, sample values and film points bit for bit; images of the same sample
numbers as for the Sobol' sampler."""
import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu
u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _pair(sc, cam, film, integ, center=False):
    r = ra.Renderer(0).configure(sc, cam, film, integ, sampler="halton", sample_at_pixel_center=center)
    o = orc.Oracle(sc, cam, film, integ, sampler="halton", sample_at_pixel_center=center)
    return r, o


@pytest.mark.parametrize("w,h,filt,radius,center", [(16, 16, "box", 0.5, False), (200, 150, "box", 0.5, False), (37, 5, "gaussian", 2.0, False),
                                                    (1, 1, "box", 0.5, False), (300, 2, "triangle", 1.5, False), (640, 360, "mitchell", 2.0, True)])
def test_halton_sample_values_and_camera_rays_bit_exact(w, h, filt, radius, center):
    lens = 2.0 if center else 0.0
    sc, cam = scenes.cornell_box("microfacet"), scenes.cornell_camera(w, h, lens_radius=lens, focal_distance=900.0)
    film = scenes.film_desc(w, h, filter_kind=filt, radius=(radius, radius))
    integ = scenes.integrator_desc(film, 100)
    r, o = _pair(sc, cam, film, integ, center)
    sb = scenes.sample_bounds(film)
    rng = np.random.RandomState(w + h)
    n = 3000
    P = np.stack([rng.randint(sb[0], sb[2], n), rng.randint(sb[1], sb[3], n)], axis=1).astype(np.int32)
    P[0] = (sb[0], sb[1]); P[1] = (sb[2] - 1, sb[3] - 1)
    S = rng.randint(0, 100, n).astype(np.int64); S[:3] = (0, 1, 99)
    n_dims = 40
    io, do = o.sobol(P, S, n_dims)
    ig, dg = r.sobol(P, S, n_dims)
    assert np.array_equal(io, ig)
    assert np.array_equal(u32(do), u32(dg))
    if center:
        assert np.all(dg[:, :2] == 0.5)
    ro, po = o.camera_rays(P, S)
    rg, pg = r.camera_rays(P, S)
    assert np.array_equal(u32(po), u32(pg))
    if lens == 0.0:
        assert np.array_equal(u32(ro), u32(rg))
    else:  # the lens point goes through sinf / cosf of two libms
        assert np.allclose(ro, rg, rtol=2e-6, atol=1e-4)
    r.close(); o.close()


def test_halton_high_sample_numbers_and_dimensions():
    """Sample numbers near the 32-bit index limit of this build and the dimensions a deep path reaches."""
    w, h = 320, 200
    sc, cam, film = scenes.cornell_box("microfacet"), scenes.cornell_camera(w, h), scenes.film_desc(w, h)
    integ = scenes.integrator_desc(film, 130000, max_depth=60)
    r, o = _pair(sc, cam, film, integ)
    rng = np.random.RandomState(9)
    n = 2000
    P = np.stack([rng.randint(0, w, n), rng.randint(0, h, n)], axis=1).astype(np.int32)
    S = rng.randint(0, 130000, n).astype(np.int64); S[:2] = (129999, 65536)
    n_dims = 5 + 8 * 60
    io, do = o.sobol(P, S, n_dims)
    ig, dg = r.sobol(P, S, n_dims)
    assert np.array_equal(io, ig) and np.array_equal(u32(do), u32(dg))
    assert float(dg.max()) < 1.0 and float(dg.min()) >= 0.0
    r.close(); o.close()


@pytest.mark.parametrize("variant,integrator,spp", [("microfacet", "path", 12), ("uber", "path", 16), ("microfacet", "direct-all", 6), ("specular", "direct-one", 16),
                                                    ("fog", "volpath", 10), ("textured", "path", 8)])
def test_halton_images(variant, integrator, spp):
    sc = scenes.cornell_fog("microfacet") if variant == "fog" else scenes.cornell_textured(True) if variant == "textured" else scenes.cornell_box(variant)
    if integrator == "direct-all":
        for l in sc.lights:
            l.n_samples = 3
    cam, film = scenes.cornell_camera(96, 80), scenes.film_desc(96, 80, filter_kind="gaussian" if variant == "microfacet" else "box")
    desc = {"path": scenes.integrator_desc(film, spp), "direct-all": scenes.direct_lighting_desc(film, spp, max_depth=4, strategy=abi.DL_SAMPLE_ALL),
            "direct-one": scenes.direct_lighting_desc(film, spp, max_depth=4, strategy=abi.DL_SAMPLE_ONE), "volpath": scenes.volpath_desc(film, spp)}[integrator]
    r, o = _pair(sc, cam, film, desc)
    r.render(0, spp); o.render(0, spp, threads=8)
    assert not o.read_past_arrays()
    a, b = r.image(), o.image()
    assert np.isfinite(a).all()
    assert common.mean_relative_error(a, b) < (5e-3 if integrator == "volpath" else 1e-3)
    so, st = o.stats(), r.stats()
    assert st["camera_samples"] == so["samples"]                      # spp as given, not rounded to a power of two
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, 2e-3 * so[k]), (k, st[k], so[k])
    # not the Sobol' sampler's picture, and the Sobol' sampler is what a later configuration without a selection gets back
    r.select_sampler("sobol")
    r.set_integrator(desc)
    r.clear(); r.render(0, min(spp, 8))
    o2 = orc.Oracle(sc, cam, film, desc)
    o2.render(0, min(spp, 8), threads=8)
    assert common.mean_relative_error(r.image(), o2.image()) < (5e-3 if integrator == "volpath" else 1e-3)
    r.close(); o.close(); o2.close()


def test_halton_sample_ranges_add_up():
    """Sample numbers [0, 5) then [5, 12) deposit what [0, 12) deposits (the spp-sharding of the multi-GPU path)."""
    sc, cam, film = scenes.cornell_box("microfacet"), scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integ = scenes.integrator_desc(film, 12)
    r, o = _pair(sc, cam, film, integ)
    r.render(0, 5); r.render(5, 12)
    two = r.film().copy()
    r.clear(); r.render(0, 12)
    one = r.film()
    assert np.allclose(two, one, rtol=1e-5, atol=1e-6)
    o.render(0, 12, threads=8)
    # 12 per pixel, one more where a sample sits on a pixel's edge and the box filter's inclusive footprint reaches the neighbour
    assert np.array_equal(one[..., 3], o.film_xyzw()[..., 3]) and 12.0 <= one[..., 3].min() and one[..., 3].mean() < 12.1
    r.close(); o.close()


def test_halton_limits_and_unknown_kind():
    r = ra.Renderer(0)
    with pytest.raises(ra.EnigmaError) as e:
        r._chk(r.L.enigma_sampler_select(r.h, 5, 0))
    assert e.value.code == -1
    film = scenes.film_desc(256, 256)
    r.upload_scene(scenes.cornell_box("diffuse")); r.set_camera(scenes.cornell_camera(256, 256)); r.set_film(film)
    r.select_sampler("halton")
    with pytest.raises(ra.EnigmaError) as e:  # (spp + 1) x 128 x 243 must fit 32 bits
        r.set_integrator(scenes.integrator_desc(film, 140000))
    assert e.value.code == -4 and "Halton index" in str(e.value)
    with pytest.raises(ra.EnigmaError):
        r.render(0, 1)                                                 # the failed configuration left no integrator
    r.set_integrator(scenes.integrator_desc(film, 130000))
    r.render(129990, 130000)
    wsum = r.film()[..., 3]
    assert wsum.min() >= 10.0 and wsum.mean() < 10.1
    r.close()


def test_sample_array_limits_of_both_samplers_on_a_fresh_handle():
    """Entry k of a sample array of n entries is a sample of sample number s * n + k (Enigma/Sampler.cpp:170-180): strategy
    "all" multiplies the largest sample number by the largest per-light sample count, and the first configuration of a handle
    already sees that (the DirectLightingIntegrator's configuration is finished by the first render after it)."""
    sc = scenes.cornell_box("diffuse")
    sc.lights[0].n_samples = 2048
    film = scenes.film_desc(16, 16)
    r = ra.Renderer(0)
    r.upload_scene(sc); r.set_camera(scenes.cornell_camera(16, 16)); r.set_film(film)

    def configure_and_render(spp):
        r.set_integrator(scenes.direct_lighting_desc(film, spp, strategy=abi.DL_SAMPLE_ALL))
        r.render(0, 1)
    with pytest.raises(ra.EnigmaError) as e:   # 2^30 x 2^11 sample numbers: past the 40 bits the Sobol' constants are tabulated for
        configure_and_render(1 << 30)
    assert e.value.code == -4 and "Sobol' index width" in str(e.value)
    configure_and_render(1 << 20)
    r.select_sampler("halton")
    with pytest.raises(ra.EnigmaError) as e:   # (2^20 x 2^11 + 1) x 16 x 27 does not fit 32 bits
        configure_and_render(1 << 20)
    assert e.value.code == -4 and "Halton index" in str(e.value)
    configure_and_render(4000)                 # 4000 x 2048 x 432 = 3.5e9 fits
    r.set_integrator(scenes.direct_lighting_desc(film, 5000, strategy=abi.DL_SAMPLE_ALL))
    r.select_sampler("sobol")                  # a selection after the configuration belongs to the next configuration
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 1)
    assert "Halton index" in str(e.value)
    r.select_sampler("halton")
    with pytest.raises(ra.EnigmaError):
        configure_and_render(5000)
    r.close()
