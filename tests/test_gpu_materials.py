"""GPU parity for the materials whose BSDF the device walks as a list of lobes: mirror, translucent, substrate, uber and mix
(Enigma/Material.cpp:318-328, :276-309, :424-448, :536-592, :146-163; lobes Enigma/BxDF.cpp:610-635 ScaledBxDF, :781-839
FresnelBlend, :1160-1190 LambertianTransmission), through the C ABI; the CPU oracle is the checker.  Point evaluations of
BSDF::f / Pdf / sample_f agree to libm precision and pick the same lobes; images agree as for the other materials (mean
relative error < 1 % asked, ~1e-4 seen), ray counts within a few paths."""
import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _directions(n, seed):
    rng = np.random.default_rng(seed)
    def sphere(k):
        v = rng.normal(size=(k, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        v[np.abs(v[:, 2]) < 0.05, 2] = 0.3          # grazing directions amplify last-bit differences of the lobes' cosines
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return sphere(n), sphere(n), rng.random((n, 2), dtype=np.float32)


def _tables():
    """name -> (materials, index of the one under test)"""
    copper = abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, 0.1)
    plastic = abi.plastic((0.3, 0.3, 0.6), (0.4, 0.4, 0.4), 0.05)
    t = {
        "mirror": ([abi.mirror((0.9, 0.8, 0.7))], 0),
        "mirror_black": ([abi.mirror((0, 0, 0))], 0),
        "translucent": ([abi.translucent((0.5, 0.45, 0.3), (0.3, 0.3, 0.3), (0.5, 0.5, 0.5), (0.6, 0.5, 0.4), 0.2)], 0),
        "translucent_reflect_only": ([abi.translucent((0.5, 0.45, 0.3), (0.3, 0.3, 0.3), (0.5, 0.5, 0.5), (0, 0, 0), 0.2)], 0),
        "translucent_no_gloss": ([abi.translucent((0.5, 0.45, 0.3), (0, 0, 0), (0.5, 0.5, 0.5), (0.6, 0.5, 0.4), 0.2, remap=False)], 0),
        "substrate": ([abi.substrate((0.5, 0.4, 0.3), (0.1, 0.1, 0.1), 0.05, 0.2)], 0),
        "substrate_no_remap": ([abi.substrate((0.5, 0.4, 0.3), (0.3, 0.2, 0.1), 0.3, 0.3, remap=False)], 0),
        "uber": ([abi.uber((0.25, 0.3, 0.6), (0.3, 0.3, 0.3), (0.2, 0.2, 0.2), (0.3, 0.3, 0.3), (0.8, 0.8, 0.8), roughness=0.05, index=1.4)], 0),
        "uber_opaque": ([abi.uber((0.25, 0.3, 0.6), (0.3, 0.3, 0.3), uroughness=0.1, vroughness=0.3)], 0),
        "uber_clear": ([abi.uber(opacity=(0, 0, 0))], 0),
        "mix_plastic_metal": ([plastic, copper, abi.mix(0, 1, (0.4, 0.4, 0.4))], 2),
        "mix_glasses_matte": ([abi.glass(1.5, 0, 0), abi.glass(1.4, 0.1, 0.05), abi.mix(0, 1), abi.matte((0.6, 0.1, 0.1), sigma=20.0),
                               abi.mix(2, 3, (0.7, 0.5, 0.9))], 4),
        "mix_of_new_kinds": ([abi.substrate(), abi.uber(kr=(0.3, 0.3, 0.3), opacity=(0.5, 0.6, 0.7)), abi.mix(0, 1, (0.3, 0.3, 0.3)), abi.mirror(),
                              abi.mix(2, 3, (0.9, 0.9, 0.9))], 4),
        # the table entry itself is one of the five kinds the shade kernels are specialised for: the tap of a table reaches them too
        "plain_plastic": ([plastic], 0),
        "plain_smooth_glass": ([abi.glass(1.5, 0, 0)], 0),
    }
    return t


def test_bsdf_point_evaluations_of_the_lobe_list_materials():
    """BSDF::f, Pdf and sample_f with BSDF_ALL of every new material, 4096 (wo, wi, u) triples each, with and without
    allowMultipleLobes: the same lobe is sampled and the values agree to libm precision."""
    wo, wi, u = _directions(4096, 0xB5DF)
    r = ra.Renderer(0)
    for name, (mats, which) in _tables().items():
        for multi in (True, False):
            got = r.bsdf_eval_of(mats, which, wo, wi, u, multiple_lobes=multi)
            want = orc.bsdf_eval_of(mats, which, wo, wi, u, multiple_lobes=multi)
            assert np.isfinite(got).all(), name
            same = got[:, 11] == want[:, 11]
            # a Fresnel-weighted choice between reflection and refraction (FresnelSpecular, Enigma/BxDF.cpp:1130) can fall
            # either way when u[0] sits within rounding of F
            assert same.mean() > 0.999, (name, multi, same.mean())
            scale = np.maximum(np.abs(want[same, :11]), 1e-5)
            err = np.abs(got[same, :11] - want[same, :11]) / scale
            assert np.quantile(err, 0.999) < 2e-3 and np.median(err) < 1e-6, (name, multi, err.max(), np.quantile(err, 0.999))
    r.close()


def test_the_tap_refuses_what_the_library_does_not_render():
    r = ra.Renderer(0)
    wo, wi, u = _directions(4, 1)
    with pytest.raises(ra.EnigmaError) as e:
        r.bsdf_eval(abi.mix(0, 0), wo, wi, u)                       # a mix alone names nothing
    assert e.value.code == -1
    with pytest.raises(ra.EnigmaError) as e:
        r.bsdf_eval_of([abi.matte((0.5,) * 3), abi.mix(0, 1)], 1, wo, wi, u)   # ... nor may it name itself
    assert e.value.code == -4
    nine = [abi.uber(kr=(0.3,) * 3, kt=(0.3,) * 3, opacity=(0.5,) * 3), abi.translucent(), abi.mix(0, 1)]   # 5 + 4 lobes
    with pytest.raises(ra.EnigmaError) as e:
        r.bsdf_eval_of(nine, 2, wo, wi, u)
    assert e.value.code == -4 and "8" in str(e.value)
    m = abi.mirror()
    m.tex_kr = 1
    with pytest.raises(ra.EnigmaError) as e:
        r.bsdf_eval(m, wo, wi, u)
    assert e.value.code == -4
    r.close()


def _compare(r, o, spp, mre=1e-3, ray_tol=1e-3):
    r.render(0, spp)
    o.render(0, spp, threads=8)
    a, b = r.image(), o.image()
    assert np.isfinite(a).all()
    err = common.mean_relative_error(a, b)
    assert err < 0.01 and err < mre, err
    so, st = o.stats(), r.stats()
    assert st["camera_samples"] == so["samples"]
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, ray_tol * so[k]), (k, st[k], so[k])
    return a


@pytest.mark.parametrize("variant", ["uber", "mix"])
def test_image_parity_path_integrator(variant):
    sc, cam, film, integ = common.cornell_setup(res=64, spp=32, variant=variant, max_depth=6)
    r, o = ra.Renderer(0).configure(sc, cam, film, integ), orc.Oracle(sc, cam, film, integ)
    img = _compare(r, o, 32)
    # the materials matter: the all-matte box differs visibly
    r2 = ra.Renderer(0).configure(scenes.cornell_box("diffuse"), cam, film, integ)
    r2.render(0, 32)
    assert common.mean_relative_error(img, r2.image()) > 0.05
    r.close(); r2.close(); o.close()


@pytest.mark.parametrize("variant", ["uber", "mix"])
def test_image_parity_in_global_memory_and_with_the_spatial_light_distribution(variant, monkeypatch):
    """The same scenes through the builds for scenes in global memory (the Cornell box otherwise runs out of LDS)."""
    monkeypatch.setenv("ENIGMA_NO_LDS_SCENE", "1")
    sc, cam, film, integ = common.cornell_setup(res=48, spp=16, variant=variant, light_strategy=abi.LIGHTS_SPATIAL)
    r, o = ra.Renderer(0).configure(sc, cam, film, integ), orc.Oracle(sc, cam, film, integ)
    _compare(r, o, 16)
    assert r.stats()["scene_in_lds"] == 0
    r.close(); o.close()


@pytest.mark.parametrize("variant", ["uber", "mix"])
def test_image_parity_volpath_in_fog(variant):
    sc = scenes.cornell_fog(variant)
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    desc = scenes.volpath_desc(film, 16, max_depth=6)
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    _compare(r, o, 16, mre=5e-3, ray_tol=2e-3)
    r.close(); o.close()


@pytest.mark.parametrize("variant", ["uber", "mix"])
def test_image_parity_direct_lighting(variant):
    """DirectLightingIntegrator: allowMultipleLobes is false (the smooth glass inside the mix is two specular lobes), and
    the recursion follows the specular lobes of the lists: the uber block's Kr and Kt and its see-through part, the mirror."""
    sc = scenes.cornell_box(variant)
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    desc = scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ONE)
    if variant == "uber":
        # Kr, Kt and opacity < 1 together: SpecularTransmit would pick one of its two lobes with a sample drawn after the whole
        # reflection subtree -- refused; without Kr the two transmission lobes are picked with the sample drawn at the hit
        r = ra.Renderer(0).configure(sc, cam, film, desc)
        with pytest.raises(ra.EnigmaError) as e:
            r.render(0, 1)
        assert e.value.code == -4
        r.close()
        ub = [i for i in range(sc.n_materials) if sc.materials[i].type == abi.MAT_UBER]
        assert len(ub) == 1
        sc.materials[ub[0]].kr[:] = (0, 0, 0)
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    _compare(r, o, 16)
    r.close(); o.close()
    if variant == "uber":
        return
    # strategy "all" with a material that both reflects and transmits specularly: refused as for a smooth glass with Kr and Kt
    desc_all = scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ALL)
    r = ra.Renderer(0).configure(sc, cam, film, desc_all)
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 1)                                      # the integrator meets the scene when the render starts
    assert e.value.code == -4
    r.close()


def test_direct_lighting_all_lights_with_a_mirror_and_a_substrate():
    """Strategy "all" where the recursion does not branch: a mirror wall (one specular lobe) and a substrate floor."""
    sc = scenes.cornell_box("diffuse")
    sc.materials[0] = abi.substrate((0.5, 0.4, 0.3), (0.2, 0.2, 0.2), 0.1, 0.1)     # white walls, floor, ceiling and blocks
    sc.materials[1] = abi.mirror((0.9, 0.9, 0.9))                                   # the left wall
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    desc = scenes.direct_lighting_desc(film, 8, max_depth=3, strategy=abi.DL_SAMPLE_ALL)
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    _compare(r, o, 8)
    r.close(); o.close()


@pytest.mark.parametrize("integrator,trilinear", [("path", True), ("path", False), ("volpath", True), ("direct", False)])
def test_image_textures_on_the_lobe_list_materials(integrator, trilinear, monkeypatch):
    """Image textures on Kd / Ks / Kr of uber, substrate, mirror, translucent and on a matte inside a mix: the lobes that
    exist at a hit follow the textures' values there (a translucent sheet has no Lambertian lobes where its Kd texture is
    black); textures without "trilinear" fold into the constants."""
    sc = scenes.cornell_textured_lobes(trilinear)
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    desc = {"path": scenes.integrator_desc(film, 16, max_depth=6), "volpath": scenes.volpath_desc(film, 16),
            "direct": scenes.direct_lighting_desc(film, 16, max_depth=3, strategy=abi.DL_SAMPLE_ONE)}[integrator]
    r, o = ra.Renderer(0).configure(sc, cam, film, desc), orc.Oracle(sc, cam, film, desc)
    img = _compare(r, o, 16, mre=2e-3, ray_tol=2e-3)
    r.close(); o.close()
    if integrator == "path" and trilinear:
        # the holes matter: with the translucent block's texture replaced by its mean the picture and the ray counts change
        monkeypatch.setenv("ENIGMA_NO_LDS_SCENE", "1")
        r = ra.Renderer(0).configure(sc, cam, film, desc)
        r.render(0, 16)
        assert common.mean_relative_error(r.image(), img) < 1e-4          # the global-memory build agrees with the LDS one
        r.close()


def test_trilinear_textures_and_specular_lobes_under_direct_lighting_are_refused_and_three_textures_too():
    sc = scenes.cornell_textured_lobes(True)
    cam, film = scenes.cornell_camera(32, 32), scenes.film_desc(32, 32)
    r = ra.Renderer(0).configure(sc, cam, film, scenes.direct_lighting_desc(film, 4, max_depth=3, strategy=abi.DL_SAMPLE_ONE))
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 1)                                          # the mirror wall's reflections would carry ray differentials on
    assert e.value.code == -4 and "trilinear" in str(e.value)
    for i in range(sc.n_materials):
        if sc.materials[i].type == abi.MAT_UBER:
            sc.materials[i].tex_kr = 4                          # a third trilinear texture on one material
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(sc)
    assert e.value.code == -4 and "two trilinear textures" in str(e.value)
    r.close()


def test_upload_refusals():
    sc = scenes.cornell_box("uber")
    r = ra.Renderer(0)
    bad = scenes.cornell_box("mix")
    for i in range(bad.n_materials):
        if bad.materials[i].type == abi.MAT_MIX:
            bad.materials[i].mix_b = bad.n_materials            # out of range and not before the mix
            break
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(bad)
    assert e.value.code == -4
    r.upload_scene(sc)                                          # the handle is still good
    r.close()
