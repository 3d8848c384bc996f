"""GPU parity of the DirectLightingIntegrator path (run with -m gpu): the HIP wavefront walk of
DirectLightingIntegrator::Li's recursion against the CPU oracle on the same scenes and sample numbers.
Same bars as the PathIntegrator tests: images within 1 % mean relative error (north-star bound), the
much tighter agreement identical samples actually give, and identical ray counts (same control flow)."""
import os

import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pair(sc, cam, film, integ):
    return ra.Renderer(0).configure(sc, cam, film, integ), orc.Oracle(sc, cam, film, integ)


def _compare(r, o, spp, threads=8, mre=1e-4, ray_tol=1e-4):
    r.render(0, spp)
    o.render(0, spp, threads=threads)
    assert not o.read_past_arrays()                      # the oracle stayed inside what the reference defines
    a, b = r.image(), o.image()
    assert np.isfinite(a).all()
    assert common.mean_relative_error(a, b) < 0.01
    assert common.mean_relative_error(a, b) < mre
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])
    sg, so = r.stats(), o.stats()
    assert sg["camera_samples"] == so["samples"]
    for kg, ko in (("rays_closest", "rays_closest"), ("rays_shadow", "rays_shadow"), ("rays_mis", "rays_mis")):
        assert abs(sg[kg] - so[ko]) <= max(3, ray_tol * so[ko]), (kg, sg[kg], so[ko])
    return a


@pytest.mark.parametrize("variant", ["diffuse", "microfacet"])
@pytest.mark.parametrize("strategy", [abi.DL_SAMPLE_ALL, abi.DL_SAMPLE_ONE])
def test_direct_lighting_cornell(variant, strategy):
    sc, cam, film, _ = common.cornell_setup(res=64, spp=16, variant=variant)
    integ = scenes.direct_lighting_desc(film, 16, max_depth=5, strategy=strategy)
    r, o = _pair(sc, cam, film, integ)
    img = _compare(r, o, 16)
    assert img.mean() > 0.01
    r.close(); o.close()


@pytest.mark.parametrize("max_depth", [1, 2, 4])
def test_direct_lighting_specular_recursion_strategy_one(max_depth):
    """Smooth glass with Kr and Kt: SpecularReflect and SpecularTransmit both spawn, the walk is a tree and the
    sampler dimensions of a transmission branch follow its sibling's whole subtree."""
    sc, cam, film, _ = common.cornell_setup(res=64, spp=8, variant="specular")
    integ = scenes.direct_lighting_desc(film, 8, max_depth=max_depth, strategy=abi.DL_SAMPLE_ONE)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 8, mre=1e-3, ray_tol=1e-3)
    if max_depth > 1:
        assert r.stats()["rays_closest"] > 64 * 64 * 8     # children were traced
    r.close(); o.close()


def test_direct_lighting_glass_single_lobe_strategy_all():
    """Kt only: one specular lobe, the recursion is a chain, so strategy 'all' stays within its sample arrays."""
    sc = scenes.cornell_box("specular")
    for m in sc.materials:
        if m.type == abi.MAT_GLASS:
            m.kr[:] = [0.0, 0.0, 0.0]
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integ = scenes.direct_lighting_desc(film, 8, max_depth=4, strategy=abi.DL_SAMPLE_ALL)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 8, mre=1e-3, ray_tol=1e-3)
    r.close(); o.close()


def test_direct_lighting_sample_arrays_with_several_samples_per_light():
    """nsamples = 4 / 2: array entries are Sobol' samples of OTHER sample numbers of the pixel."""
    sc = scenes.cornell_box("diffuse")
    sc.lights[0].n_samples = 4
    sc.lights[1].n_samples = 2
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    integ = scenes.direct_lighting_desc(film, 16, max_depth=3, strategy=abi.DL_SAMPLE_ALL)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 16)
    # sample ranges stay additive
    whole = r.film()
    r.clear(); r.render(0, 7); r.render(7, 16)
    assert np.allclose(r.film(), whole, rtol=1e-5, atol=1e-5)
    r.close(); o.close()


def test_direct_lighting_many_lights_strategy_one_and_switching_integrators():
    sc = scenes.many_lights()
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integ = scenes.direct_lighting_desc(film, 16, strategy=abi.DL_SAMPLE_ONE)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 16, mre=1e-3)
    # back to the PathIntegrator on the same handle
    pinteg = scenes.integrator_desc(film, 16)
    r.set_integrator(pinteg); o.set_integrator(pinteg)
    r.clear(); o.clear()
    r.render(0, 8); o.render(0, 8, threads=8)
    assert common.mean_relative_error(r.image(), o.image()) < 1e-3
    r.close(); o.close()


@pytest.mark.parametrize("strategy", [abi.DL_SAMPLE_ALL, abi.DL_SAMPLE_ONE])
def test_direct_lighting_spot_and_distant_lights(strategy):
    sc = scenes.cornell_box("diffuse", delta_lights=True)
    sc.lights[0].n_samples = 2
    sc.lights[2].n_samples = 3                             # the spot light: a delta light sampled three times under strategy all
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integ = scenes.direct_lighting_desc(film, 8, max_depth=3, strategy=strategy)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 8)
    r.close(); o.close()


def test_direct_lighting_refuses_what_the_reference_leaves_undefined():
    cam, film = scenes.cornell_camera(32, 32), scenes.film_desc(32, 32)
    # strategy all + branching glass: the reference indexes past its sample arrays
    r = ra.Renderer(0).configure(scenes.cornell_box("specular"), cam, film, scenes.direct_lighting_desc(film, 4, max_depth=3))
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 4)
    assert e.value.code == -4
    o = orc.Oracle(scenes.cornell_box("specular"), cam, film, scenes.direct_lighting_desc(film, 4, max_depth=3))
    o.render(0, 4, threads=2)
    assert o.read_past_arrays()                           # the oracle sees the same overrun
    o.close()
    r.set_integrator(scenes.direct_lighting_desc(film, 4, max_depth=1))   # depth 1: one Li per sample, defined
    r.render(0, 4)
    r.close()
    # strategy all over 576 lights needs 5 + 4 * 576 * 5 Sobol' dimensions
    r = ra.Renderer(0).configure(scenes.many_lights(), cam, film, scenes.direct_lighting_desc(film, 4))
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 4)
    assert e.value.code == -4
    r.close()


def test_direct_lighting_matches_golden_fixture():
    g = np.load(os.path.join(GOLD, "direct_lighting64_16spp.npz"))
    sc, cam, film, _ = common.cornell_setup(res=64, spp=16, variant="microfacet")
    sc.lights[0].n_samples = 2
    r = ra.Renderer(0).configure(sc, cam, film, scenes.direct_lighting_desc(film, 16, max_depth=5, strategy=abi.DL_SAMPLE_ALL))
    r.render(0, 16)
    assert common.mean_relative_error(r.image(), g["all_rgb"]) < 1e-4
    st = r.stats()
    got = np.array([st["camera_samples"], st["rays_closest"], st["rays_shadow"], st["rays_mis"]], dtype=np.int64)
    assert np.all(np.abs(got - g["all_rays"]) <= np.maximum(3, 1e-4 * g["all_rays"]))
    r.close()
    sc, cam, film, _ = common.cornell_setup(res=64, spp=16, variant="specular")
    r = ra.Renderer(0).configure(sc, cam, film, scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ONE))
    r.render(0, 16)
    assert common.mean_relative_error(r.image(), g["one_rgb"]) < 1e-3
    st = r.stats()
    got = np.array([st["camera_samples"], st["rays_closest"], st["rays_shadow"], st["rays_mis"]], dtype=np.int64)
    assert np.all(np.abs(got - g["one_rays"]) <= np.maximum(3, 1e-3 * g["one_rays"]))
    r.close()


@pytest.mark.parametrize("integrator,strategy", [("path", abi.LIGHTS_POWER), ("path", abi.LIGHTS_SPATIAL), ("direct-all", None), ("volpath", abi.LIGHTS_POWER)])
def test_projection_light_renders_like_the_oracle(integrator, strategy):
    """ENIGMA_LIGHT_PROJECTION: the reference's ProjectionLight never has a map (Enigma/Lights.cpp:220-221): a point light whose
    Power() is its cone's.  Next to the Cornell box's area light the power and spatial distributions must weigh it as the oracle does."""
    extra = [abi.projection_light((150.0, 400.0, 150.0), (6.0e5, 4.0e5, 3.0e5), fov=60.0, n_samples=2)]
    sc = scenes.cornell_fog("diffuse", extra_lights=extra) if integrator == "volpath" else scenes.cornell_box("microfacet", extra_lights=extra)
    assert sc.lights[sc.n_lights - 1].type == abi.LIGHT_PROJECTION
    cam, film = scenes.cornell_camera(80, 64), scenes.film_desc(80, 64)
    desc = {"path": lambda: scenes.integrator_desc(film, 16, light_strategy=strategy), "volpath": lambda: scenes.volpath_desc(film, 16, light_strategy=strategy),
            "direct-all": lambda: scenes.direct_lighting_desc(film, 16, strategy=abi.DL_SAMPLE_ALL)}[integrator]()
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    r.render(0, 16); o.render(0, 16, threads=8)
    a, b = r.image(), o.image()
    assert common.mean_relative_error(a, b) < (5e-3 if integrator == "volpath" else 1e-3)
    so, st = o.stats(), r.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, 2e-3 * so[k]), (k, st[k], so[k])
    r.close(); o.close()


@pytest.mark.parametrize("integrator,strategy,with_map", [("path", abi.LIGHTS_POWER, True), ("path", abi.LIGHTS_SPATIAL, True), ("path", abi.LIGHTS_POWER, False), ("direct-all", None, True)])
def test_goniometric_light_renders_like_the_oracle(integrator, strategy, with_map):
    """ENIGMA_LIGHT_GONIOMETRIC: a point light of intensity I x the top texel of its map's pyramid (a non-power-of-two map goes
    through the reference's Lanczos zoom first), weighed in the light distributions by 4 pi^2 x lookup((.5, .5), .5)."""
    gmap = scenes.sky_map(12, 20) if with_map else None
    extra = [abi.goniometric_light((150.0, 400.0, 150.0), (4.0e5, 4.0e5, 4.0e5), gmap, n_samples=2)]
    sc = scenes.cornell_box("microfacet", extra_lights=extra)
    cam, film = scenes.cornell_camera(80, 64), scenes.film_desc(80, 64)
    desc = scenes.integrator_desc(film, 16, light_strategy=strategy) if integrator == "path" else scenes.direct_lighting_desc(film, 16, strategy=abi.DL_SAMPLE_ALL)
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    r.render(0, 16); o.render(0, 16, threads=8)
    assert common.mean_relative_error(r.image(), o.image()) < 1e-3
    so, st = o.stats(), r.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, 2e-3 * so[k]), (k, st[k], so[k])
    r.close(); o.close()
