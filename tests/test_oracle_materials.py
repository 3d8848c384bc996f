"""Known answers for the oracle's mirror, translucent, substrate, uber and mix materials (Enigma/Material.cpp:318-328,
:276-309, :424-448, :536-592, :146-163) and the lobes only they use (Enigma/BxDF.cpp:610-635 ScaledBxDF, :781-839
FresnelBlend, :1160-1190 LambertianTransmission): closed forms of BSDF::f / Pdf / sample_f at a surface whose normal is +z.
The reference holds no vector for any of them: parity unpinned, as for the other materials."""
import math

import numpy as np

import orc
from raytrace_amd import abi, scenes

REFL_SPEC, TRANS_SPEC, REFL_DIFF, TRANS_DIFF, REFL_GLOSSY = 1 | 16, 2 | 16, 1 | 4, 2 | 4, 1 | 8


def _dirs(n, seed=7):
    rng = np.random.default_rng(seed)
    def sphere():
        v = rng.normal(size=(n, 3))
        v[:, 2] = np.where(np.abs(v[:, 2]) < 0.1, 0.5, v[:, 2])
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return sphere(), sphere(), rng.random((n, 2), dtype=np.float32)


def _up(v):
    v = v.copy()
    v[:, 2] = np.abs(v[:, 2])
    return v


def test_mirror_is_one_specular_reflection_without_fresnel():
    wo, wi, u = _dirs(64)
    kr = np.array([0.9, 0.8, 0.7], dtype=np.float32)
    out = orc.bsdf_eval(abi.mirror(tuple(kr)), wo, wi, u)
    assert not out[:, :4].any()                                          # f and Pdf of a specular lobe are 0
    assert np.array_equal(out[:, 8:11], wo * np.array([-1, -1, 1], dtype=np.float32))
    assert np.array_equal(out[:, 7], np.ones(64, dtype=np.float32)) and np.array_equal(out[:, 11], np.full(64, REFL_SPEC, dtype=np.float32))
    assert np.allclose(out[:, 4:7] * np.abs(wo[:, 2:3]), kr, rtol=1e-6)  # f |cos| / pdf = Kr: FresnelNoOp
    black = orc.bsdf_eval(abi.mirror((0, 0, 0)), wo, wi, u)
    assert not black.any()                                               # no lobe at all


def test_translucent_without_gloss_is_two_lambertian_lobes_one_per_side():
    wo, wi, u = _dirs(256)
    kd, r, t = np.array([0.5, 0.4, 0.3]), np.array([0.5, 0.6, 0.7]), np.array([0.2, 0.3, 0.4])
    m = abi.translucent(tuple(kd), (0, 0, 0), tuple(r), tuple(t), 0.2)
    out = orc.bsdf_eval(m, wo, wi, u)
    same = wo[:, 2] * wi[:, 2] > 0
    assert np.allclose(out[same, :3], r * kd / math.pi, rtol=1e-6) and np.allclose(out[~same, :3], t * kd / math.pi, rtol=1e-6)
    assert np.allclose(out[:, 3], np.abs(wi[:, 2]) / (2 * math.pi), rtol=1e-6)   # each lobe's density lives on its side; two lobes
    # u[0] < 1/2 picks the reflection lobe, the rest the transmission lobe, which leaves on the far side
    first = u[:, 0] < 0.5
    assert np.array_equal(out[:, 11], np.where(first, REFL_DIFF, TRANS_DIFF).astype(np.float32))
    assert np.all(np.sign(out[first, 10]) == np.sign(wo[first, 2])) and np.all(np.sign(out[~first, 10]) == -np.sign(wo[~first, 2]))
    # transmit = 0 leaves the reflection lobes only; reflect = transmit = 0 leaves nothing, whatever Kd and Ks are
    refl = orc.bsdf_eval(abi.translucent(tuple(kd), (0.3,) * 3, tuple(r), (0, 0, 0), 0.2), wo, wi, u)
    assert not refl[~same, :3].any() and refl[same, :3].min() > 0
    assert not orc.bsdf_eval(abi.translucent(tuple(kd), (0.3,) * 3, (0, 0, 0), (0, 0, 0), 0.2), wo, wi, u).any()


def test_substrate_fresnel_blend_closed_form_at_normal_incidence_and_energy():
    rd, rs, a = np.array([0.5, 0.4, 0.3]), np.array([0.1, 0.2, 0.3]), 0.25
    m = abi.substrate(tuple(rd), tuple(rs), a, a, remap=False)
    z = np.array([[0, 0, 1]], dtype=np.float32)
    out = orc.bsdf_eval(m, z, z, np.array([[0.3, 0.3]], dtype=np.float32))[0]
    diffuse = 28 / (23 * math.pi) * rd * (1 - rs) * (1 - 0.5 ** 5) ** 2
    specular = (1 / (math.pi * a * a)) / 4 * rs                          # D(n) / (4 |wi.wh| max(cos, cos)) * Schlick(1) = Rs
    assert np.allclose(out[:3], diffuse + specular, rtol=1e-5)
    assert np.isclose(out[3], 0.5 * (1 / math.pi + (1 / (math.pi * a * a)) / 4), rtol=1e-5)   # half cosine, half visible-normal density
    # sampling: u[0] < 1/2 draws the cosine lobe, else the microfacet normal; every sample stays on wo's side
    wo, wi, u = _dirs(8192, 11)
    wo = _up(wo)
    s = orc.bsdf_eval(abi.substrate((0.5,) * 3, (0.5,) * 3, 0.1, 0.3), wo, wi, u)
    ok = s[:, 7] > 0
    assert ok.mean() > 0.8 and np.all(s[ok, 10] > 0) and np.all(s[ok, 11] == REFL_GLOSSY)
    albedo = np.where(ok[:, None], s[:, 4:7] * np.abs(s[:, 10:11]) / np.maximum(s[:, 7:8], 1e-30), 0).mean(axis=0)
    assert np.all(albedo < 1.0) and np.all(albedo > 0.3)                 # reflects less than it receives, and not little
    assert not orc.bsdf_eval(abi.substrate((0,) * 3, (0,) * 3), wo, wi, u).any()


def test_uber_lobes_follow_opacity_and_the_four_colours():
    wo, wi, u = _dirs(512, 5)
    # fully transparent: one specular transmission with both indices 1 -- the ray goes straight on with weight 1
    out = orc.bsdf_eval(abi.uber(opacity=(0, 0, 0)), wo, wi, u)
    assert np.allclose(out[:, 8:11], -wo, atol=1e-6) and np.all(out[:, 11] == TRANS_SPEC) and np.all(out[:, 7] == 1)   # Refract's own rounding
    assert np.allclose(out[:, 4:7] * np.abs(wo[:, 2:3]), 1.0, rtol=1e-5)
    # opaque, Ks = Kr = Kt = 0: the matte material
    kd = (0.25, 0.3, 0.6)
    assert np.array_equal(orc.bsdf_eval(abi.uber(kd, (0, 0, 0)), wo, wi, u), orc.bsdf_eval(abi.matte(kd), wo, wi, u))
    # opaque with Kd and Ks: a plastic whose Fresnel term faces the other way round (1 -> index instead of 1.5 -> 1); same lobes, same densities
    ub, pl = orc.bsdf_eval(abi.uber(kd, (0.4,) * 3, roughness=0.1, index=1.5), wo, wi, u), orc.bsdf_eval(abi.plastic(kd, (0.4,) * 3, 0.1), wo, wi, u)
    assert np.array_equal(ub[:, 3], pl[:, 3]) and np.array_equal(ub[:, 11], pl[:, 11]) and not np.allclose(ub[:, :3], pl[:, :3])
    # all five lobes, in the order they are added: see-through, Lambert, glossy, mirror, glass; u[0] picks one in five
    m = abi.uber(kd, (0.3,) * 3, (0.2,) * 3, (0.3,) * 3, (0.8,) * 3, roughness=0.05, index=1.4)
    out = orc.bsdf_eval(m, wo, wi, u)
    order = np.array([TRANS_SPEC, REFL_DIFF, REFL_GLOSSY, REFL_SPEC, TRANS_SPEC], dtype=np.float32)
    picked = np.minimum((u[:, 0] * 5).astype(int), 4)
    ok = out[:, 7] > 0
    assert ok.mean() > 0.8 and np.array_equal(out[ok, 11], order[picked[ok]])
    see = ok & (picked == 0)
    assert np.allclose(out[see, 8:11], -wo[see], atol=1e-6)              # the see-through part does not bend the ray
    assert np.allclose(out[see, 4:7] * np.abs(wo[see, 2:3]), 0.2, rtol=1e-5) and np.allclose(out[see, 7], 0.2)   # pdf 1 / five lobes
    bent = ok & (picked == 4)
    assert not np.allclose(out[bent, 8:11], -wo[bent], atol=1e-3)       # ... the glass part does (index 1.4)
    # uroughness / vroughness as the constructor resolves them: u alone sets both
    a = abi.uber(kd, (0.3,) * 3, uroughness=0.2)
    assert a.uroughness == a.vroughness == np.float32(0.2)


def test_mix_scales_the_lobes_and_answers_densities_with_the_cosine_hemisphere():
    wo, wi, u = _dirs(512, 9)
    copper = abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, 0.2)
    matte = abi.matte((0.6, 0.5, 0.4))
    table = [copper, matte, abi.mix(0, 1, (0.25, 0.5, 0.75))]
    fm, fd, fx = orc.bsdf_eval(copper, wo, wi, u), orc.bsdf_eval(matte, wo, wi, u), orc.bsdf_eval_of(table, 2, wo, wi, u)
    s1 = np.array([0.25, 0.5, 0.75])
    assert np.allclose(fx[:, :3], s1 * fm[:, :3] + (1 - s1) * fd[:, :3], rtol=1e-5, atol=1e-7)
    # ScaledBxDF has no pdf() of its own (Enigma/BxDF.h:157-170): both lobes answer with the base class's density, so the
    # mixture's Pdf is the cosine density whatever the metal's own is
    same = wo[:, 2] * wi[:, 2] > 0
    assert np.allclose(fx[same, 3], np.abs(wi[same, 2]) / math.pi, rtol=1e-6) and not fx[~same, 3].any()
    assert not np.allclose(fm[same, 3], np.abs(wi[same, 2]) / math.pi, rtol=1e-2)
    # a sample of the metal lobe carries the metal's own density for itself plus the matte lobe's cosine density, halved
    first = (u[:, 0] < 0.5) & (fx[:, 7] > 0)
    um = u.copy(); um[:, 0] = np.minimum(u[:, 0] * 2, np.float32(1 - 2 ** -24))
    alone = orc.bsdf_eval(copper, wo, wi, um)
    assert np.allclose(fx[first, 8:11], alone[first, 8:11], atol=1e-6)
    assert np.allclose(fx[first, 7], 0.5 * (alone[first, 7] + np.abs(alone[first, 10]) / math.pi), rtol=1e-5)
    # amounts are clamped below only (Spectrum::Clamp(0, inf)): 1.5 weighs the first material 1.5 and the second 0
    over = orc.bsdf_eval_of([copper, matte, abi.mix(0, 1, (1.5, 1.5, 1.5))], 2, wo, wi, u)
    assert np.allclose(over[:, :3], 1.5 * fm[:, :3], rtol=1e-6)
    # a mix of a mix multiplies the weights; a material mixed with itself is itself
    nested = orc.bsdf_eval_of([copper, matte, abi.mix(0, 1, (0.5,) * 3), abi.mix(2, 1, (0.5,) * 3)], 3, wo, wi, u)
    assert np.allclose(nested[:, :3], 0.25 * fm[:, :3] + 0.25 * fd[:, :3] + 0.5 * fd[:, :3], rtol=1e-5, atol=1e-7)
    twice = orc.bsdf_eval_of([matte, abi.mix(0, 0, (0.5,) * 3)], 1, wo, wi, u)
    assert np.allclose(twice[:, :3], fd[:, :3], rtol=1e-6)


def test_smooth_glass_inside_a_mix_follows_allow_multiple_lobes():
    """PathIntegrator: one FresnelSpecular lobe (reflection or refraction by u[0] < F); DirectLightingIntegrator: a
    specular reflection and a specular transmission lobe (Enigma/Material.cpp:191-218), chosen by halves of u[0]."""
    wo, wi, u = _dirs(512, 13)
    table = [abi.glass(1.5, 0, 0), abi.matte((0, 0, 0)), abi.mix(0, 1, (1, 1, 1))]   # the black matte adds no lobe
    multi = orc.bsdf_eval_of(table, 2, wo, wi, u, multiple_lobes=True)
    single = orc.bsdf_eval_of(table, 2, wo, wi, u, multiple_lobes=False)
    plain = orc.bsdf_eval(abi.glass(1.5, 0, 0), wo, wi, u)
    assert np.array_equal(np.delete(multi, 3, axis=1), np.delete(plain, 3, axis=1))
    # ... but for BSDF::Pdf: the wrapped specular lobe would say 0, the wrapper says cosine hemisphere (ScaledBxDF has no pdf())
    same = wo[:, 2] * wi[:, 2] > 0
    assert not plain[:, 3].any() and np.allclose(multi[same, 3], np.abs(wi[same, 2]) / math.pi, rtol=1e-6) and not multi[~same, 3].any()
    ok = single[:, 7] > 0
    assert np.array_equal(single[ok, 11], np.where(u[ok, 0] < 0.5, REFL_SPEC, TRANS_SPEC).astype(np.float32))
    assert np.allclose(single[ok, 7], 0.5)
    lone = orc.bsdf_eval_of([abi.glass(1.5, 0, 0)], 0, wo, wi, u, multiple_lobes=False)
    assert np.array_equal(np.delete(lone, 3, axis=1), np.delete(single, 3, axis=1))


def test_scenes_with_the_new_materials_render_and_differ_from_the_matte_box():
    cam, film = scenes.cornell_camera(24, 24), scenes.film_desc(24, 24)
    base = None
    for variant in ("diffuse", "uber", "mix"):
        o = orc.Oracle(scenes.cornell_box(variant), cam, film, scenes.integrator_desc(film, 8))
        o.render(0, 8, threads=4)
        img = o.image()
        o.close()
        assert np.isfinite(img).all() and img.min() >= 0 and img.mean() > 0.01
        if base is None:
            base = img
        else:
            assert np.mean(np.abs(img - base)) > 0.02 * base.mean()
    bad = scenes.cornell_box("mix")
    for i in range(bad.n_materials):
        if bad.materials[i].type == abi.MAT_MIX:
            bad.materials[i].mix_a = i
            break
    try:
        orc.Oracle(bad, cam, film, scenes.integrator_desc(film, 8))
        raise AssertionError("a mix that names itself was accepted")
    except RuntimeError:
        pass
