"""GPU parity for the VolPathIntegrator (Enigma/VolPathIntegrator.cpp:45-177) with homogeneous media
(Enigma/Medium.cpp:69-108) and the Henyey-Greenstein phase function, through the C ABI; the CPU oracle is the checker.
Tolerances as for the other integrators: mean relative error < 1 % (north star), in practice ~1e-4: same samples, same
arithmetic up to expf / logf / sinf / cosf and the order in which a contribution's factors are multiplied."""
import os

import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes
from test_oracle_kat import _slab_scene

pytestmark = pytest.mark.gpu


def _pair(sc, cam, film, integ):
    return ra.Renderer(0).configure(sc, cam, film, integ), orc.Oracle(sc, cam, film, integ)


def _compare(r, o, spp, mre=1e-3, ray_tol=2e-3):
    r.render(0, spp)
    o.render(0, spp, threads=8)
    a, b = r.image(), o.image()
    assert np.isfinite(a).all()
    assert common.mean_relative_error(a, b) < 0.01
    assert common.mean_relative_error(a, b) < mre
    so, st = o.stats(), r.stats()
    assert st["camera_samples"] == so["samples"]
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, ray_tol * so[k]), (k, st[k], so[k])


def test_volpath_without_media_matches_the_path_integrator_on_the_gpu():
    sc, cam, film, integ = common.cornell_setup(res=48, spp=8, variant="diffuse")
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(0, 8)
    path_img, path_stats = r.image(), r.stats()
    r.set_integrator(scenes.volpath_desc(film, 8))
    r.clear()
    r.render(0, 8)
    assert np.allclose(r.image(), path_img, rtol=1e-6, atol=1e-7)
    st = r.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert st[k] == 2 * path_stats[k]                    # the counters run on across renders
    r.close()


def test_volpath_ends_paths_whose_throughput_is_black():
    """VolPathIntegrator::Li tests beta.IsBlack() after every walk, in a medium or not (Enigma/VolPathIntegrator.cpp:71),
    the PathIntegrator never does: between walls of pure red, green and blue (and a glass that reflects red and transmits
    blue) a path's throughput multiplies to black, and from there on the two integrators trace different numbers of rays for the same picture.  Counts must
    match the oracle's for both."""
    sc = scenes.cornell_box("specular")
    pure = [[0.8, 0.0, 0.0], [0.0, 0.0, 0.8], [0.0, 0.8, 0.0]]           # walls and floor red, the side walls blue and green
    for i in range(sc.n_materials):
        m = sc.materials[i]
        if m.type == abi.MAT_GLASS:
            m.kr[:] = [1.0, 0.0, 0.0]
            m.kt[:] = [0.0, 0.0, 1.0]
        elif m.type == abi.MAT_MATTE:
            m.kd[:] = pure[i % 3]
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    counts = {}
    for name, desc in (("volpath", scenes.volpath_desc(film, 8, max_depth=6)), ("path", scenes.integrator_desc(film, 8, max_depth=6))):
        r, o = _pair(sc, cam, film, desc)
        r.render(0, 8); o.render(0, 8, threads=4)
        sg, so = r.stats(), o.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(sg[k] - so[k]) <= max(2, so[k] // 10000), (name, k, sg[k], so[k])   # a libm-level flip or two; the rule itself is worth 5 %
        assert common.mean_relative_error(r.image(), o.image()) < 1e-4
        counts[name] = sg["rays_closest"]
        r.close(); o.close()
    assert counts["volpath"] < 0.97 * counts["path"]      # the test of the throughput is what ends those paths early


@pytest.mark.parametrize("variant", ["diffuse", "microfacet", "specular"])
def test_volpath_fog_cornell(variant):
    """Fog slab (scattering, g = 0.3) in front of the blocks, absorbing medium inside the tall block: medium sampling,
    phase-function light sampling + MIS, transmittance along shadow and MIS rays through material-less boundaries."""
    sc = scenes.cornell_fog(variant)
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    r, o = _pair(sc, cam, film, scenes.volpath_desc(film, 16))
    _compare(r, o, 16, mre=1e-3 if variant == "diffuse" else 5e-3)
    # the fog changes the picture: same scene without media differs visibly
    r2 = ra.Renderer(0).configure(scenes.cornell_box(variant), cam, film, scenes.volpath_desc(film, 16))
    r2.render(0, 16)
    assert common.mean_relative_error(r.image(), r2.image()) > 0.02
    r.close(); r2.close(); o.close()


def test_volpath_uniform_light_sampling_back_scattering_and_deep_paths():
    sc = scenes.cornell_fog("diffuse", g=-0.5)
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    integ = scenes.volpath_desc(film, 8, max_depth=12, light_strategy=abi.LIGHTS_UNIFORM)
    r, o = _pair(sc, cam, film, integ)
    _compare(r, o, 8)
    r.close(); o.close()


def test_volpath_absorbing_slab_beer_lambert_on_the_gpu():
    sigma, d = 0.25, 4.0
    sc = _slab_scene((sigma, sigma, sigma), (0.0, 0.0, 0.0), thickness=d)
    cam = scenes.camera_desc((0, 0, -20), (0, 0, 0), (0, 1, 0), 4.0, 8, 8)
    film = scenes.film_desc(8, 8)
    r = ra.Renderer(0).configure(sc, cam, film, scenes.volpath_desc(film, 1024, max_depth=3))
    r.render(0, 1024)
    want = np.array([1.0, 2.0, 3.0]) * np.exp(-sigma * d)
    assert np.allclose(r.image().astype(np.float64).mean(axis=(0, 1)), want, rtol=0.01)
    # the PathIntegrator skips the slab's faces and knows nothing of its medium: the wall arrives unattenuated, three rays a
    # sample (face, face, wall; the black wall ends the path)
    r.set_integrator(scenes.integrator_desc(film, 4, max_depth=3))
    r.clear()
    before = r.stats()["rays_closest"]
    r.render(0, 4)
    assert np.allclose(r.image().astype(np.float64).mean(axis=(0, 1)), [1.0, 2.0, 3.0], rtol=1e-5)
    assert r.stats()["rays_closest"] - before == 3 * 4 * 64
    # ... and so does the DirectLightingIntegrator: Li goes on behind the faces (Enigma/DirectLightingIntegrator.cpp:51-54)
    r.set_integrator(scenes.direct_lighting_desc(film, 4, max_depth=3))
    r.clear()
    r.render(0, 4)
    assert np.allclose(r.image().astype(np.float64).mean(axis=(0, 1)), [1.0, 2.0, 3.0], rtol=1e-5)
    r.close()


def test_grid_density_medium_is_black_like_the_references_on_the_gpu():
    """MakeMedium "heterogeneous" makes the reference's GridDensityMedium, whose Tr and Sample return black
    (Enigma/Medium.cpp:51-59).  A slab of it before an emissive wall: black behind it, the wall beside it; then the Cornell
    box in a fog of that kind, where paths end at their first segment in the medium and no light crosses it: images and ray
    counts of library and oracle, the LDS build and the build for scenes in global memory."""
    sc = _slab_scene((0.01, 0.01, 0.01), (0.0, 0.0, 0.0), thickness=4.0)
    sc.media[0].kind = abi.MEDIUM_GRID_DENSITY
    film = scenes.film_desc(32, 32)
    wide = scenes.camera_desc((0, 0, -20), (0, 0, 0), (0, 1, 0), 120.0, 32, 32)       # the slab in the middle, the wall around it
    integ = scenes.volpath_desc(film, 16, max_depth=3)
    r = ra.Renderer(0).configure(sc, wide, film, integ); o = orc.Oracle(sc, wide, film, integ)
    r.render(0, 16); o.render(0, 16, threads=4)
    a, b = r.image(), o.image()
    assert np.all(a[14:18, 14:18] == 0) and a[0, 0].sum() > 0
    assert common.mean_relative_error(a, b) < 1e-5
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert r.stats()[k] == o.stats()[k], k
    r.close(); o.close()
    for no_lds in (False, True):
        fog = scenes.cornell_fog("microfacet")
        for m in fog.media[:fog.n_media]:
            m.kind = abi.MEDIUM_GRID_DENSITY
        cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
        integ = scenes.volpath_desc(film, 16)
        if no_lds:
            os.environ["ENIGMA_NO_LDS_SCENE"] = "1"
        try:
            r = ra.Renderer(0).configure(fog, cam, film, integ)
            r.render(0, 8)
        finally:
            os.environ.pop("ENIGMA_NO_LDS_SCENE", None)
        o = orc.Oracle(fog, cam, film, integ)
        o.render(0, 8, threads=8)
        assert r.stats()["scene_in_lds"] == (0 if no_lds else 1)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-4
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(r.stats()[k] - o.stats()[k]) <= 2, (k, r.stats()[k], o.stats()[k])
        r.close(); o.close()


def test_volpath_delta_lights_through_fog_and_sample_ranges_add_up():
    """Spot and distant lights seen through the fog (VisibilityTester::Tr towards a point / a far point)."""
    base = scenes.cornell_fog("diffuse")
    extra = [abi.spot_light((278, 500, 100), (186, 165, 169), (4.0e5, 5.0e5, 6.0e5), 25.0, 10.0), abi.distant_light((0.3, 0.5, -1.1), (0, 0, 0), (1.5, 1.2, 0.9))]
    lights = [base.lights[i] for i in range(base.n_lights)] + extra
    sc = abi.SceneArrays(base.vertices, base.indices, base.tri_material, base.tri_light, [base.materials[i] for i in range(base.n_materials)], lights,
                         tri_flags=base.tri_flags, media=[base.media[i] for i in range(base.n_media)],
                         tri_medium_inside=base.tri_medium_inside, tri_medium_outside=base.tri_medium_outside)
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    r, o = _pair(sc, cam, film, scenes.volpath_desc(film, 16))
    _compare(r, o, 16)
    whole = r.film()
    r.clear(); r.render(0, 5); r.render(5, 16)
    assert np.allclose(r.film(), whole, rtol=1e-5, atol=1e-5)
    r.close(); o.close()


def test_volpath_full_size_1080p_invariants_and_window():
    """The VolPathIntegrator at the BASELINE size (1920 x 1080 sampler and film): film invariants of one pass over the
    fog scene, additivity of sample ranges, and a window of the same job against the oracle pixel by pixel."""
    sc = scenes.cornell_fog("microfacet")
    cam = scenes.cornell_camera(1920, 1080)
    film = scenes.film_desc(1920, 1080)
    integ = scenes.volpath_desc(film, 512)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(508, 512)
    f = r.film()
    st = r.stats()
    assert st["camera_samples"] == 1920 * 1080 * 4
    assert f[..., 3].min() >= 4 and f[..., 3].sum() - 4 * 1920 * 1080 < 0.01 * 4 * 1920 * 1080
    assert np.isfinite(f).all() and f[..., :3].min() >= 0
    whole = ra.resolve(f)
    r.clear(); r.render(508, 510); r.render(510, 512)
    assert np.allclose(ra.resolve(r.film()), whole, rtol=1e-4, atol=1e-5)
    integ_w = scenes.volpath_desc(film, 512)
    integ_w.pixel_bounds[:] = [700, 700, 892, 808]
    r.set_integrator(integ_w)
    r.clear(); r.render(508, 512)
    o = orc.Oracle(sc, cam, film, integ_w)
    o.render(508, 512, threads=16)
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])
    win = (slice(700, 808), slice(700, 892))
    a, b = r.image()[win], o.image()[win]
    assert np.allclose(whole[win][1:-1, 1:-1], a[1:-1, 1:-1], rtol=1e-4, atol=1e-5)
    assert common.mean_relative_error(a, b) < 5e-3
    r.close(); o.close()


def test_volpath_textured_big_mesh_in_fog_full_size_1080p():
    """The useful part of BASELINE configs[4] at its full size: the 1.18 M-triangle mesh with uvs and trilinear image
    textures inside a homogeneous medium, VolPathIntegrator, 1920 x 1080, 512-spp sampler (the reference's heterogeneous
    medium returns black from both of its methods, Enigma/Medium.cpp:51-64; the test after this one covers it).  Film invariants over the whole film, additivity
    of sample ranges, a window of the same job against the oracle pixel by pixel, ray counts, and the window's primary
    hits out of the closest-hit kernel that renders."""
    sc = scenes.big_mesh_fog()
    assert sc.indices.shape[0] >= 1 << 20 and sc.n_textures == 3 and sc.n_media == 2
    cam = scenes.big_mesh_camera(1920, 1080)
    film = scenes.film_desc(1920, 1080)
    integ = scenes.volpath_desc(film, 512)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(510, 512)
    f = r.film()
    st = r.stats()
    assert st["camera_samples"] == 1920 * 1080 * 2 and st["stack_overflows"] == 0
    assert f[..., 3].min() >= 2 and f[..., 3].sum() - 2 * 1920 * 1080 < 0.01 * 2 * 1920 * 1080
    assert np.isfinite(f).all() and f[..., :3].min() >= 0
    assert (f[..., :3].sum(axis=2) > 0).mean() > 0.9            # in-scattered light nearly everywhere, at two samples a pixel
    whole = ra.resolve(f)
    r.clear(); r.render(510, 511); r.render(511, 512)
    assert np.allclose(ra.resolve(r.film()), whole, rtol=1e-4, atol=1e-5)
    integ_w = scenes.volpath_desc(film, 512)
    integ_w.pixel_bounds[:] = [600, 600, 792, 708]
    r.set_integrator(integ_w)
    r.clear()
    before = r.stats()
    r.render(510, 512)
    after = r.stats()
    o = orc.Oracle(sc, cam, film, integ_w)
    o.render(510, 512, threads=16)
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])
    win = (slice(600, 708), slice(600, 792))
    a, b = r.image()[win], o.image()[win]
    assert np.allclose(whole[win][1:-1, 1:-1], a[1:-1, 1:-1], rtol=1e-4, atol=1e-5)   # the window is a piece of the whole job
    assert common.mean_relative_error(a, b) < 0.01                                  # the north star's bound
    assert np.median(np.abs(a - b) / np.maximum(b, 1e-3)) < 1e-5                    # what is reached: the same paths
    so = o.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(after[k] - before[k]) - int(so[k])) <= 2e-3 * so[k] + 2, k
    # tier 1 on the kernel that renders (the VolPathIntegrator's build of it)
    import test_gpu_parity
    test_gpu_parity._primary_tap_check(r, o, (600, 600, 792, 708), (0, 511))
    r.close(); o.close()


def test_volpath_big_mesh_in_the_references_heterogeneous_medium_1080p_window():
    """BASELINE configs[4] as it is worded -- a heterogeneous medium -- and as the reference defines such a medium: its
    GridDensityMedium returns black from Tr and Sample (Enigma/Medium.cpp:51-59).  The textured 1.18 M-triangle mesh with its
    fog box filled by that medium, 1080p, 512-spp sampler: a window of the job against the oracle, ray for ray.  Paths end at
    their first segment inside the box, so only what the camera sees before the box, and nothing lit through it, remains."""
    sc = scenes.big_mesh_fog()
    for m in sc.media[:sc.n_media]:
        m.kind = abi.MEDIUM_GRID_DENSITY
    cam = scenes.big_mesh_camera(1920, 1080)
    film = scenes.film_desc(1920, 1080)
    integ = scenes.volpath_desc(film, 512)
    integ.pixel_bounds[:] = [600, 600, 792, 708]
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(510, 512); o.render(510, 512, threads=16)
    win = (slice(600, 708), slice(600, 792))
    a, b = r.image()[win], o.image()[win]
    assert np.array_equal(r.film()[..., 3], o.film_xyzw()[..., 3])
    assert np.isfinite(a).all() and np.allclose(a, b, rtol=1e-4, atol=1e-6)
    sg, so = r.stats(), o.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sg[k]) - int(so[k])) <= 2e-3 * so[k] + 2, (k, sg[k], so[k])
    assert sg["rays_closest"] <= 2.1 * sg["camera_samples"]     # the box's face, then one segment inside: the path is over
    r.close(); o.close()
