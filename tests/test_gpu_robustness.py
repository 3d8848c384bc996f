"""Error behaviour of the C ABI on a GPU box: a call that fails leaves the handle in a state that the next call
reports, never in a mixture of two configurations; bogus sizes come back as error codes, not as aborts."""
import copy

import numpy as np
import pytest

import common
import raytrace_amd as ra
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu

E_INVALID, E_STATE = -1, -2


def _renderer():
    sc, cam, film, integ = common.cornell_setup(res=32, spp=4, variant="microfacet")
    return ra.Renderer(0).configure(sc, cam, film, integ), sc, cam, film, integ


def test_failed_scene_upload_leaves_no_scene():
    r, sc, cam, film, integ = _renderer()
    r.render(0, 1)
    bad = scenes.cornell_box("diffuse")
    bad.tri_light = bad.tri_light.copy()
    bad.tri_light[-1] = 1000  # light index out of range: found after the material tables were rebuilt
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(bad)
    assert e.value.code == E_INVALID
    with pytest.raises(ra.EnigmaError) as e:
        r.render(1, 2)
    assert e.value.code == E_STATE
    with pytest.raises(ra.EnigmaError) as e:
        r.intersect(np.array([[278, 273, -800, 0, 0, 1, np.inf]], dtype=np.float32))
    assert e.value.code == E_STATE
    r.upload_scene(sc)   # a good upload makes the handle usable again
    r.render(1, 2)
    assert np.isfinite(r.image()).all()
    r.close()


def test_rejected_integrator_config_keeps_the_old_one():
    """Arguments refused before anything is rewritten leave the configured integrator usable; the handle never
    holds a half-rebuilt sampler (a failure after the rewrite has begun clears the integrator: haveInteg)."""
    r, sc, cam, film, integ = _renderer()
    r.render(0, 1)
    before = r.film()
    bad = copy.copy(integ)
    bad.samples_per_pixel = 0
    with pytest.raises(ra.EnigmaError) as e:
        r.set_integrator(bad)
    assert e.value.code == E_INVALID
    r.clear()
    r.render(0, 1)   # still the 4-spp integrator of the set-up
    assert np.array_equal(r.film(), before)
    r.close()


@pytest.mark.parametrize("bounds", [(0, 0, 1 << 30, 1 << 30), (0, 0, 70000, 8), (-(1 << 31), 0, (1 << 31) - 1, 8), (5, 5, 5, 9)])
def test_bogus_sample_bounds_are_an_error_code(bounds):
    r, sc, cam, film, integ = _renderer()
    bad = copy.copy(integ)
    bad.sample_bounds[:] = bounds
    with pytest.raises(ra.EnigmaError) as e:
        r.set_integrator(bad)
    assert e.value.code == E_INVALID
    r.close()


@pytest.mark.parametrize("value", [float("nan"), float("inf"), -float("inf"), 3.0e38, -2.0e19])
def test_coordinates_that_are_not_finite_or_out_of_range_are_an_error_code(value):
    """A NaN, an infinity or a coordinate beyond 2^60 would reach the tree builder (bins, surface areas) and the traversal
    arithmetic: the upload refuses it, a valid upload afterwards works, 2^60 itself is accepted."""
    import numpy as np
    sc = scenes.cornell_box("diffuse")
    good = sc.vertices.copy()
    r = ra.Renderer(0)
    sc.vertices[7, 1] = value
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(sc)
    assert e.value.code == E_INVALID and "vertex 7" in str(e.value)
    sc.vertices[:] = good
    sc.vertices[7, 1] = np.float32(2.0 ** 60)
    r.upload_scene(sc)
    # a sphere whose radius or transform is not finite
    sp = scenes.cornell_spheres()
    sp.spheres[0].object_to_world[3] = value
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(sp)
    assert e.value.code == E_INVALID and "sphere 0" in str(e.value)
    r.close()


def test_sample_count_beyond_the_sobol_index_width_is_refused():
    # resolution 4096 leaves 52 - 24 = 28 bits for the sample number (Enigma/SamplerInclude.h:2600-2637)
    sc = scenes.cornell_box("diffuse")
    film = scenes.film_desc(4096, 8, "box")
    ok = scenes.integrator_desc(film, 1 << 28)
    bad = scenes.integrator_desc(film, 1 << 29)
    r = ra.Renderer(0)
    r.upload_scene(sc); r.set_camera(scenes.cornell_camera(4096, 8)); r.set_film(film)
    r.set_integrator(ok)
    with pytest.raises(ra.EnigmaError) as e:
        r.set_integrator(bad)
    assert e.value.code == E_INVALID
    r.close()


def test_sphere_media_without_per_triangle_arrays_are_kept():
    """A scene whose media hang on spheres only, with triangles that carry no interface arrays: the spheres'
    interfaces must act (the image differs from the same scene without the medium)."""
    def build(with_medium):
        base = scenes.cornell_box("diffuse")
        m = np.eye(4); m[:3, 3] = (278, 200, 278)
        media = [abi.homogeneous_medium((0.02, 0.02, 0.02), (0.0, 0.0, 0.0))] if with_medium else []
        ball = abi.sphere(abi.NO_MATERIAL, radius=120.0, object_to_world=m, medium_inside=0 if with_medium else -1)
        return abi.SceneArrays(base.vertices, base.indices, base.tri_material, base.tri_light, list(base.materials)[:base.n_materials],
                               list(base.lights)[:base.n_lights], media=media, spheres=[ball])
    film = scenes.film_desc(32, 32, "box")
    integ = scenes.volpath_desc(film, 16)
    imgs = []
    for with_medium in (False, True):
        r = ra.Renderer(0).configure(build(with_medium), scenes.cornell_camera(32, 32), film, integ)
        r.render(0, 16)
        imgs.append(r.image())
        r.close()
    centre = (slice(12, 20), slice(12, 20))
    assert imgs[1][centre].mean() < 0.9 * imgs[0][centre].mean()
