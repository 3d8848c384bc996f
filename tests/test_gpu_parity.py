"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI
(raytrace_amd/libenigma_hip.so via ctypes); the CPU oracle is only the checker.

Tier 1 (bit-exact): Sobol sample values, camera rays (pinhole), closest-hit primitive ids / t /
barycentrics, any-hit decisions, film footprint pixel indices and weights.
Tier 2 (floating point): images.  Tolerance: mean relative error < 1 % and every pixel within
3 sigma of the Monte Carlo estimator (the north-star bounds); the tests also assert the much
tighter agreement actually achieved (same samples, same arithmetic up to libm and sum order).
"""
import os

import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def pair(setup):
    sc, cam, film, integ = setup
    return ra.Renderer(0).configure(sc, cam, film, integ), orc.Oracle(sc, cam, film, integ)


# ------------------------------------------------------------------ tier 1
@pytest.mark.parametrize("xr,yr,spp", [(512, 512, 16), (1920, 1080, 1024), (37, 23, 4), (1, 1, 4)])
def test_sobol_values_bit_exact(xr, yr, spp):
    r, o = pair(common.cornell_setup(xres=xr, yres=yr, spp=spp))
    rng = np.random.RandomState(5)
    n = 4000
    P = np.stack([rng.randint(0, xr, n), rng.randint(0, yr, n)], axis=1).astype(np.int32)
    P[:4] = [(0, 0), (xr - 1, yr - 1), (xr - 1, 0), (0, yr - 1)]
    S = rng.randint(0, spp, n).astype(np.int64)
    S[:8] = [0, spp - 1, 0, spp - 1, 1, 2, spp // 2, 0]   # first and last sample numbers
    io, do = o.sobol(P, S, 45)
    ig, dg = r.sobol(P, S, 45)
    assert np.array_equal(io, ig)
    assert np.array_equal(u32(do), u32(dg))
    r.close(); o.close()


def test_sobol_and_camera_match_golden_fixture():
    g = np.load(os.path.join(GOLD, "sobol_camera.npz"))
    for name, (xr, yr, spp) in {"m9": (512, 512, 16), "m11": (1920, 1080, 1024)}.items():
        sc, cam, film, integ = common.cornell_setup(xres=xr, yres=yr, spp=spp)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        idx, dims = r.sobol(g[name + "_pixels"], g[name + "_samples"], 8)
        rays, pf = r.camera_rays(g[name + "_pixels"], g[name + "_samples"])
        assert np.array_equal(idx, g[name + "_index"])
        assert np.array_equal(u32(dims), g[name + "_dims_bits"])
        assert np.array_equal(u32(rays), g[name + "_rays_bits"])
        assert np.array_equal(u32(pf), g[name + "_pfilm_bits"])
        r.close()


def test_camera_rays_bit_exact_pinhole_and_close_with_lens():
    r, o = pair(common.cornell_setup(xres=256, yres=144, spp=64))
    rng = np.random.RandomState(9)
    P = np.stack([rng.randint(0, 256, 3000), rng.randint(0, 144, 3000)], axis=1).astype(np.int32)
    S = rng.randint(0, 64, 3000).astype(np.int64)
    ro, po = o.camera_rays(P, S)
    rg, pg = r.camera_rays(P, S)
    assert np.array_equal(u32(ro), u32(rg)) and np.array_equal(u32(po), u32(pg))
    r.close(); o.close()
    # thin lens: ConcentricSampleDisk goes through cos/sin, whose last bit differs between libm and the device
    r, o = pair(common.cornell_setup(xres=256, yres=144, spp=64, lens_radius=8.0, focal_distance=1000.0))
    ro, po = o.camera_rays(P, S)
    rg, pg = r.camera_rays(P, S)
    assert np.array_equal(u32(po), u32(pg))
    assert np.allclose(ro, rg, rtol=2e-6, atol=1e-4)
    r.close(); o.close()


def _hit_check(r, o, rays7, allow_ambiguous=True):
    ho = o.intersect(rays7)
    hg = r.intersect(rays7)
    ok = ho["ambiguous"] == 0 if allow_ambiguous else np.ones(len(ho["tri"]), dtype=bool)
    assert np.array_equal(ho["prim_id"][ok], hg["prim_id"][ok])
    assert np.array_equal(ho["tri"][ok], hg["tri"][ok])
    assert np.array_equal(u32(ho["t"][ok]), u32(hg["t"][ok]))
    assert np.array_equal(u32(ho["bary"][ok]), u32(hg["bary"][ok]))
    # ambiguous rays (exact ties on shared edges / coplanar geometry): same distance to within the tie window
    amb = ~ok
    if amb.any():
        assert np.allclose(ho["t"][amb], hg["t"][amb], rtol=2e-6)
    return ho, int(amb.sum())


def _primary_tap_check(r, o, region, sample_nums, n_triangles=None):
    """enigma_debug_primary_hits -- k_generate + the closest-hit kernel of a real pass, hit records read out of the path
    state -- against the oracle's Scene::intersect of the same camera rays, bit for bit.  region = x0, y0, x1, y1 of the
    render region.  Returns the number of rays compared and of rays that hit."""
    x0, y0, x1, y1 = region
    ys, xs = np.mgrid[y0:y1, x0:x1]
    P = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
    n_hit = 0
    for sn in sample_nums:
        rays, _ = o.camera_rays(P, np.full(P.shape[0], sn, dtype=np.int64))
        ho = o.intersect(common.rays_from_camera(rays))
        hg = r.primary_hits(sn, (x1 - x0, y1 - y0))
        ok = ho["ambiguous"] == 0
        assert ok.mean() > 0.99
        assert np.array_equal(ho["prim_id"][ok], hg["prim_id"].ravel()[ok])
        assert np.array_equal(ho["tri"][ok], hg["tri"].ravel()[ok])
        assert np.array_equal(u32(ho["t"][ok]), u32(hg["t"].ravel()[ok]))
        on_tri = ok if n_triangles is None else ok & (ho["tri"] < n_triangles)   # a sphere hit carries its object-space point there
        assert np.array_equal(u32(ho["bary"][on_tri]), u32(hg["bary"].reshape(-1, 3)[on_tri]))
        assert np.allclose(ho["t"][~ok], hg["t"].ravel()[~ok], rtol=2e-6)   # exact ties on shared edges: same distance
        n_hit += int((ho["tri"] >= 0).sum())
    return P.shape[0] * len(sample_nums), n_hit


def test_primary_hits_of_the_production_kernel_cornell_512():
    """BASELINE configs[0] at its own size: the hit of every camera ray of sample numbers 0, 7 and 15 out of the kernel
    that renders (scene staged into LDS, ray pool, voted steps), for the three integrators' builds of it."""
    sc, cam, film, integ = common.cornell_setup(res=512, spp=16, variant="diffuse")
    r, o = pair((sc, cam, film, integ))
    n, hits = _primary_tap_check(r, o, (0, 0, 512, 512), (0, 7, 15))
    assert hits > 0.9 * n
    # the tap leaves the film alone and a render after it is unchanged
    r.render(0, 4)
    a = r.film()
    r.clear(); r.primary_hits(3, (512, 512)); r.render(0, 4)
    assert np.array_equal(a, r.film())
    for other in (scenes.direct_lighting_desc(film, 16, strategy=abi.DL_SAMPLE_ONE), scenes.volpath_desc(film, 16)):
        r.set_integrator(other); o.set_integrator(other)
        _primary_tap_check(r, o, (0, 0, 512, 512), (5,))
    with pytest.raises(ra.EnigmaError):
        r.primary_hits(16, (512, 512))            # sample number outside the sampler's range
    with pytest.raises(ra.EnigmaError):
        r.primary_hits(0, (511, 512))             # arrays of the wrong size
    r.close(); o.close()


def test_primary_hits_of_the_production_kernel_hbm_builds_and_spheres():
    """The same tap on the builds that walk the tree in HBM: a mesh too large for LDS, and a scene with spheres."""
    film = scenes.film_desc(96, 64)
    integ = scenes.integrator_desc(film, 16)
    for sc, cam in ((scenes.big_mesh(grid=96, sphere_level=4, n_spheres=4), scenes.big_mesh_camera(96, 64)),
                    (scenes.cornell_spheres(), scenes.cornell_camera(96, 64))):
        r, o = pair((sc, cam, film, integ))
        n, hits = _primary_tap_check(r, o, (0, 0, 96, 64), (0, 9), n_triangles=sc.indices.shape[0])
        assert hits > 0.5 * n
        r.close(); o.close()


@pytest.mark.parametrize("variant", ["diffuse", "microfacet"])
def test_primary_hits_bit_exact_cornell(variant):
    r, o = pair(common.cornell_setup(res=512, spp=16, variant=variant))
    ys, xs = np.mgrid[0:512:8, 0:512:8]
    P = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
    total_amb = 0
    for sn in (0, 1, 15):
        rays, _ = o.camera_rays(P, np.full(P.shape[0], sn, dtype=np.int64))
        ho, amb = _hit_check(r, o, common.rays_from_camera(rays))
        total_amb += amb
        assert (ho["tri"] >= 0).mean() > 0.5
    assert total_amb < 0.01 * 3 * P.shape[0]
    r.close(); o.close()


def test_primary_hits_match_golden_fixture():
    g = np.load(os.path.join(GOLD, "primary_hits.npz"))
    sc, cam, film, integ = common.cornell_setup(res=64, spp=16)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    for sn in (0, 7):
        rays = g["diffuse_s%d_rays_bits" % sn].view(np.float32)
        h = r.intersect(common.rays_from_camera(rays))
        ok = g["diffuse_s%d_ambiguous" % sn] == 0
        assert np.array_equal(h["prim_id"][ok], g["diffuse_s%d_prim_id" % sn][ok])
        assert np.array_equal(u32(h["t"])[ok], g["diffuse_s%d_t_bits" % sn][ok])
        assert np.array_equal(u32(h["bary"])[ok], g["diffuse_s%d_bary_bits" % sn][ok])
    r.close()


def test_hits_bit_exact_random_rays_many_lights_and_mesh():
    rng = np.random.RandomState(21)
    for sc, cam in ((scenes.many_lights(), scenes.cornell_camera(64, 64)),
                    (scenes.big_mesh(grid=96, sphere_level=4, n_spheres=4), scenes.big_mesh_camera(64, 64))):
        film = scenes.film_desc(64, 64)
        integ = scenes.integrator_desc(film, 16)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        P = np.stack([rng.randint(0, 64, 1500), rng.randint(0, 64, 1500)], axis=1).astype(np.int32)
        rays, _ = o.camera_rays(P, rng.randint(0, 16, 1500).astype(np.int64))
        ho, _ = _hit_check(r, o, common.rays_from_camera(rays))
        # secondary rays: from hit points in random directions, finite tMax too
        hit = ho["tri"] >= 0
        org = rays[hit, :3] + rays[hit, 3:6] * (ho["t"][hit, None] * 0.999)
        d = rng.normal(size=org.shape).astype(np.float32)
        sec = np.concatenate([org, d, rng.uniform(0.5, 400.0, size=(org.shape[0], 1))], axis=1).astype(np.float32)
        _hit_check(r, o, sec)
        assert np.array_equal(o.intersect_any(sec), r.intersect_any(sec))
        r.close(); o.close()


def test_traversal_stack_spill_path_gives_identical_hits():
    """Deep trees keep 12 stack levels in LDS and spill the rest to scratch.  With only two LDS
    levels (debug knob) nearly every ray spills; hits must not change by a bit."""
    rng = np.random.RandomState(33)
    sc, cam = scenes.big_mesh(grid=96, sphere_level=4, n_spheres=4), scenes.big_mesh_camera(64, 64)
    film = scenes.film_desc(64, 64)
    integ = scenes.integrator_desc(film, 16)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    P = np.stack([rng.randint(0, 64, 2000), rng.randint(0, 64, 2000)], axis=1).astype(np.int32)
    rays, _ = o.camera_rays(P, rng.randint(0, 16, 2000).astype(np.int64))
    rays7 = common.rays_from_camera(rays)
    base = r.intersect(rays7)
    hit = base["tri"] >= 0
    org = rays[hit, :3] + rays[hit, 3:6] * (base["t"][hit, None] * 0.999)
    sec = np.concatenate([org, rng.normal(size=org.shape), rng.uniform(0.5, 400.0, size=(org.shape[0], 1))], axis=1).astype(np.float32)
    base_sec, base_any = r.intersect(sec), r.intersect_any(sec)
    os.environ["ENIGMA_DEBUG_TINY_STACK"] = "1"
    try:
        tiny, tiny_sec, tiny_any = r.intersect(rays7), r.intersect(sec), r.intersect_any(sec)
    finally:
        del os.environ["ENIGMA_DEBUG_TINY_STACK"]
    for a, b in ((base, tiny), (base_sec, tiny_sec)):
        assert np.array_equal(a["tri"], b["tri"]) and np.array_equal(u32(a["t"]), u32(b["t"])) and np.array_equal(u32(a["bary"]), u32(b["bary"]))
    assert np.array_equal(base_any, tiny_any)
    _hit_check(r, o, sec)
    r.close(); o.close()


def test_degenerate_rays():
    r, o = pair(common.cornell_setup(res=32, spp=4))
    rays = np.array([
        [278, 273, -800, 0, 0, 1, np.inf],      # axis-aligned: zero direction components (inf reciprocals)
        [278, 273, 100, 0, 1, 0, np.inf],
        [278, 273, 100, -1, 0, 0, np.inf],
        [278, 273, 100, 0, -1, 0, 50.0],        # tMax shorter than the floor distance
        [278, 273, 100, 0, 0, 1, 0.0],          # tMax = 0
        [10000, 10000, 10000, 1, 1, 1, np.inf], # misses everything
        [278, 548.0, 279, 0, -1, 0, np.inf],    # origin on the light plane
    ], dtype=np.float32)
    _hit_check(r, o, rays)
    assert np.array_equal(o.intersect_any(rays), r.intersect_any(rays))
    r.close(); o.close()


@pytest.mark.parametrize("kind", ["box", "triangle", "gaussian", "mitchell", "sinc"])
def test_film_footprint_bit_exact(kind):
    g = np.load(os.path.join(GOLD, "film_footprints.npz"))
    sc, cam, film, integ = common.cornell_setup(res=64, spp=16, filter_kind=kind)
    assert np.array_equal(u32(np.array(film.filter_table[:], dtype=np.float32)), g[kind + "_table_bits"])
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    pf = g["pfilm_bits"].view(np.float32)
    pix, w, cnt = r.film_footprint(pf, 32)
    assert np.array_equal(cnt, g[kind + "_cnt"])
    for i in range(pf.shape[0]):
        c = min(int(cnt[i]), 32)
        assert np.array_equal(pix[i, :c], g[kind + "_pix"][i, :c])
        assert np.array_equal(u32(w[i, :c]), g[kind + "_w_bits"][i, :c])
    # and live against the oracle on fresh positions, including exact pixel corners
    o = orc.Oracle(sc, cam, film, integ)
    rng = np.random.RandomState(4)
    pf2 = np.concatenate([rng.uniform(-3, 67, size=(500, 2)), rng.randint(0, 64, size=(100, 2)).astype(np.float64)]).astype(np.float32)
    a, b = o.film_footprint(pf2, 32), r.film_footprint(pf2, 32)
    assert np.array_equal(a[2], b[2])
    for i in range(pf2.shape[0]):
        c = min(int(a[2][i]), 32)
        assert np.array_equal(a[0][i, :c], b[0][i, :c]) and np.array_equal(u32(a[1][i, :c]), u32(b[1][i, :c]))
    r.close(); o.close()


# ------------------------------------------------------------------ tier 2
def _sigma_check(img, ref_rgb, moments, spp):
    n = np.maximum(moments[..., 6:7], 1)
    mean = moments[..., 0:3] / n
    var = np.maximum(moments[..., 3:6] / n - mean * mean, 0)
    sigma = np.sqrt(var / n)                              # standard error of the pixel estimate
    out = np.abs(img - ref_rgb) > 3 * sigma + 1e-4 * np.maximum(ref_rgb, 1e-3)
    return float(out.mean())


@pytest.mark.parametrize("variant", ["diffuse", "microfacet", "specular"])
def test_image_parity_cornell(variant):
    g = np.load(os.path.join(GOLD, "cornell64_64spp.npz"))
    sc, cam, film, integ = common.cornell_setup(res=64, spp=64, variant=variant)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(0, 64)
    img = r.image()
    ref = g[variant + "_rgb"]
    assert np.isfinite(img).all()
    mre = common.mean_relative_error(img, ref)
    assert mre < 0.01                                       # north-star tolerance: 1 % mean relative error
    assert _sigma_check(img, ref, g[variant + "_moments"].astype(np.float64), 64) == 0.0   # every pixel within 3 sigma
    assert mre < 1e-4                                       # what identical samples + arithmetic actually give
    st = r.stats()
    want = g[variant + "_rays"]
    got = np.array([st["camera_samples"], st["rays_closest"], st["rays_shadow"], st["rays_mis"]], dtype=np.int64)
    assert got[0] == want[0]
    assert np.all(np.abs(got[1:] - want[1:]) <= np.maximum(3, 1e-4 * want[1:]))   # same control flow, path by path
    r.close()


@pytest.mark.parametrize("kind,variant", [("gaussian", "diffuse"), ("triangle", "microfacet"), ("mitchell", "diffuse"), ("sinc", "diffuse")])
def test_image_parity_wide_filters(kind, variant):
    setup = common.cornell_setup(res=48, spp=16, variant=variant, filter_kind=kind, crop=(0.1, 0.9, 0.1, 0.9))
    r, o = pair(setup)
    r.render(0, 16)
    o.render(0, 16, threads=4)
    assert common.mean_relative_error(r.image(), o.image()) < 1e-4
    f = r.film()
    assert np.allclose(f[..., 3], o.film_xyzw()[..., 3], rtol=1e-5, atol=1e-5)   # filter weight sums
    r.close(); o.close()


def test_image_parity_many_lights_power_and_uniform():
    sc = scenes.many_lights()
    for strat in (abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM):
        film = scenes.film_desc(64, 64)
        integ = scenes.integrator_desc(film, 16, light_strategy=strat)
        cam = scenes.cornell_camera(64, 64)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        r.close(); o.close()


@pytest.mark.parametrize("variant,strategy", [("diffuse", abi.LIGHTS_POWER), ("microfacet", abi.LIGHTS_UNIFORM)])
def test_image_parity_spot_and_distant_lights(variant, strategy):
    """SpotLight / DistantLight (Enigma/Lights.cpp:133-190,656-708) beside the area light; the distant light's
    direction is not unit length, as the reference leaves it."""
    g = np.load(os.path.join(GOLD, "delta_lights64_16spp.npz"))
    sc = scenes.cornell_box(variant, delta_lights=True)
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integ = scenes.integrator_desc(film, 16, light_strategy=strategy)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(0, 16)
    o.render(0, 16, threads=8)
    img = r.image()
    assert common.mean_relative_error(img, o.image()) < (1e-4 if variant == "diffuse" else 1e-3)
    assert common.mean_relative_error(img, g[variant + "_rgb"]) < (1e-4 if variant == "diffuse" else 1e-3)
    so, st = o.stats(), r.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(st[k] - so[k]) <= max(3, 1e-3 * so[k])
    # the delta lights add light: the same scene without them is darker everywhere it matters
    r2 = ra.Renderer(0).configure(scenes.cornell_box(variant), cam, film, integ)
    r2.render(0, 16)
    assert img.mean() > 1.2 * r2.image().mean()
    r.close(); r2.close(); o.close()


@pytest.mark.parametrize("radiance_map", [False, True])
def test_image_parity_infinite_area_light(radiance_map):
    """InfiniteAreaLight (Enigma/Lights.cpp:395-546): constant environment and a latitude-longitude map, under the three
    integrators; escaped camera / specular rays read the map, light samples come from its Distribution2D, BSDF samples
    that escape are weighted with the reference's Pdf_Li."""
    sc = scenes.cornell_sky("microfacet", radiance_map)
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    for integ in (scenes.integrator_desc(film, 16), scenes.integrator_desc(film, 16, light_strategy=abi.LIGHTS_UNIFORM),
                  scenes.direct_lighting_desc(film, 16, max_depth=3, strategy=abi.DL_SAMPLE_ONE), scenes.volpath_desc(film, 16)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 1e-3 * so[k])
        r.close(); o.close()
    # a map whose sides are not powers of two is resampled first (MipMap::resizeImage, Enigma/MipMap.h:155-237)
    odd = abi.infinite_light((1, 1, 1), scenes.sky_map(12, 20), light_to_world=[[1, 0, 0], [0, 0, 1], [0, -1, 0]])
    base = scenes.cornell_box("diffuse")
    sc2 = abi.SceneArrays(base.vertices, base.indices, base.tri_material, base.tri_light, [base.materials[i] for i in range(base.n_materials)],
                          [base.lights[0], base.lights[1], odd], tri_flags=base.tri_flags)
    integ = scenes.integrator_desc(film, 16)
    r = ra.Renderer(0).configure(sc2, cam, film, integ)
    o = orc.Oracle(sc2, cam, film, integ)
    r.render(0, 16); o.render(0, 16, threads=8)
    assert common.mean_relative_error(r.image(), o.image()) < 1e-3
    r.close(); o.close()


def test_image_parity_shading_normals_and_uvs():
    """Per-vertex normals and uvs (Enigma/TriangleMesh.cpp:238-262,290-371; BSDF frame Enigma/BxDF.cpp:843-852): four
    coarse spheres -- anisotropic metal with normals + uvs, rough glass with normals, plastic with uvs only, smooth
    glass with normals and ReverseOrientation -- under the three integrators; primary hits stay bit-exact."""
    sc = scenes.cornell_smooth()
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    for integ in (scenes.integrator_desc(film, 16), scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ONE),
                  scenes.volpath_desc(film, 16)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert np.isfinite(r.image()).all()
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 1e-3 * so[k])
        r.close(); o.close()
    # the shading data changes the picture: the same spheres without normals / uvs look faceted
    flat = abi.SceneArrays(sc.vertices, sc.indices, sc.tri_material, sc.tri_light, [sc.materials[i] for i in range(sc.n_materials)],
                           [sc.lights[i] for i in range(sc.n_lights)], tri_flags=sc.tri_flags & abi.TRI_FLIP_NORMAL)
    integ = scenes.integrator_desc(film, 16)
    a, b = ra.Renderer(0).configure(sc, cam, film, integ), ra.Renderer(0).configure(flat, cam, film, integ)
    a.render(0, 16); b.render(0, 16)
    assert common.mean_relative_error(a.image(), b.image()) > 0.02
    a.close(); b.close()
    # flags without their arrays, and an emissive triangle whose normals (zero here) pick no side of its plane, are refused
    bad = abi.SceneArrays(sc.vertices, sc.indices, sc.tri_material, sc.tri_light, [sc.materials[i] for i in range(sc.n_materials)],
                          [sc.lights[i] for i in range(sc.n_lights)], tri_flags=sc.tri_flags)
    with pytest.raises(ra.EnigmaError) as e:
        ra.Renderer(0).configure(bad, cam, film, integ)
    assert e.value.code == -1
    fl = sc.tri_flags.copy()
    fl[sc.tri_light >= 0] |= abi.TRI_HAS_NORMALS
    bad = abi.SceneArrays(sc.vertices, sc.indices, sc.tri_material, sc.tri_light, [sc.materials[i] for i in range(sc.n_materials)],
                          [sc.lights[i] for i in range(sc.n_lights)], tri_flags=fl, normals=sc.normals, uvs=sc.uvs)
    with pytest.raises(ra.EnigmaError) as e:
        ra.Renderer(0).configure(bad, cam, film, integ)
    assert e.value.code == -4


@pytest.mark.parametrize("normal,reverse", [((0.0, -1.0, 0.0), False), ((0.3, 1.0, 0.2), False), ((0.3, 1.0, 0.2), True)])
def test_image_parity_emissive_triangles_with_normals(normal, reverse):
    """A DiffuseAreaLight on a mesh with normals: Triangle::sample turns the sampled normal to the vertex normals
    (Enigma/TriangleMesh.cpp:618-623), Triangle::intersect turns the hit normal to shading.n, which ReverseOrientation
    flips (:369, Enigma/Interaction.cpp:139-141).  The one-sided panel of the oracle's known-answer test under the
    three integrators and the spatial light distribution; normals that straddle the panel's plane are refused."""
    from test_oracle_kat import _panel_scene
    sc = _panel_scene(normal=normal, reverse=reverse)
    cam, film = scenes.camera_desc((0, 60, 0), (0, 0, 0), (0, 0, 1), 40.0, 32, 32), scenes.film_desc(32, 32)
    for integ in (scenes.integrator_desc(film, 64), scenes.integrator_desc(film, 64, light_strategy=abi.LIGHTS_SPATIAL),
                  scenes.direct_lighting_desc(film, 64, max_depth=1, strategy=abi.DL_SAMPLE_ALL), scenes.volpath_desc(film, 64)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 64)
        o.render(0, 64, threads=8)
        assert np.isfinite(r.image()).all()
        assert o.image().max() > 1e-3         # every variant shows its light somewhere (the last only through BSDF sampling)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):   # the panel's own top samples points in its own plane: a cosine of +-1 ulp decides those few rays
            assert abs(st[k] - so[k]) <= max(16, 1e-3 * so[k])
        r.close(); o.close()
    if reverse:
        return
    b = scenes.MeshBuilder()
    m = b.add_material(abi.matte((0.5, 0.5, 0.5)))
    b.add_tri((0, 0, 0), (1, 0, 0), (0, 0, 1), m, emit=(1.0, 1.0, 1.0), normals=((0, 1, 0), (0, 1, 0), (0, -1, 0)))
    with pytest.raises(ra.EnigmaError) as e:
        ra.Renderer(0).configure(b.arrays(), cam, film, scenes.integrator_desc(film, 4))
    assert e.value.code == -4 and "normals" in str(e.value)


@pytest.mark.parametrize("trilinear,wrap", [(True, abi.WRAP_REPEAT), (True, abi.WRAP_CLAMP), (True, abi.WRAP_BLACK), (False, abi.WRAP_REPEAT)])
def test_image_parity_image_textures(trilinear, wrap):
    """Image textures on Kd / Ks / Kt / eta (Enigma/Texture.h:313-415, Enigma/MipMap.h:309-376, Enigma/Interaction.cpp:161-209):
    camera rays filtered through their ray differentials, later rays at level 0; textures without "trilinear" as their
    average colour, gamma false as ones; under the three integrators, with and without a lens."""
    sc = scenes.cornell_textured(trilinear, wrap)
    film = scenes.film_desc(64, 64)
    cases = [(scenes.cornell_camera(64, 64), scenes.integrator_desc(film, 16)),
             (scenes.cornell_camera(64, 64, lens_radius=12.0, focal_distance=900.0), scenes.integrator_desc(film, 16)),
             (scenes.cornell_camera(64, 64), scenes.direct_lighting_desc(film, 16, max_depth=3, strategy=abi.DL_SAMPLE_ALL)),
             (scenes.cornell_camera(64, 64), scenes.volpath_desc(film, 16))]
    for cam, integ in cases if wrap == abi.WRAP_REPEAT else cases[:1]:
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert np.isfinite(r.image()).all()
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 1e-3 * so[k])
        r.close(); o.close()


def test_textures_change_the_picture_and_bad_ones_are_refused():
    film, cam = scenes.film_desc(48, 48), scenes.cornell_camera(48, 48)
    integ = scenes.integrator_desc(film, 8)
    imgs = []
    for sc in (scenes.cornell_textured(True), scenes.cornell_textured(False)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        r.render(0, 8)
        imgs.append(r.image())
        r.close()
    assert common.mean_relative_error(imgs[0], imgs[1]) > 0.02          # trilinear shows the pattern, the default shows its average
    sc = scenes.cornell_textured(True)
    base = dict(vertices=sc.vertices, indices=sc.indices, tri_material=sc.tri_material, tri_light=sc.tri_light,
                materials=[sc.materials[i] for i in range(sc.n_materials)], lights=[sc.lights[i] for i in range(sc.n_lights)],
                tri_flags=sc.tri_flags, uvs=sc.uvs)
    # sides that are not powers of two are resampled first (MipMap::resizeImage), under each wrap mode
    for wrap in (abi.WRAP_REPEAT, abi.WRAP_CLAMP, abi.WRAP_BLACK):
        odd = abi.SceneArrays(**base, textures=[abi.image_texture(scenes.test_pattern(24, 40), trilinear=True, wrap=wrap, su=2.0, sv=2.0)] * sc.n_textures)
        r = ra.Renderer(0).configure(odd, cam, film, integ)
        o = orc.Oracle(odd, cam, film, integ)
        r.render(0, 8); o.render(0, 8, threads=8)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        r.close(); o.close()
    missing = abi.SceneArrays(**base, textures=[sc.textures[0]])       # the materials index five textures
    with pytest.raises(ra.EnigmaError) as e:
        ra.Renderer(0).configure(missing, cam, film, integ)
    assert e.value.code == -1
    # trilinear textures + smooth glass under the DirectLightingIntegrator: its specular rays would carry differentials
    b = scenes.MeshBuilder()
    m = b.add_material(abi.matte((0, 0, 0)))
    b.materials[m].tex_kd = b.add_texture(abi.image_texture(scenes.test_pattern(), trilinear=True))
    g = b.add_material(abi.glass(1.5, 0.0, 0.0))
    b.add_quad([(-50, 0, -50), (-50, 0, 50), (50, 0, 50), (50, 0, -50)], m, toward=(0, 1, 0), uvs=[(0, 0), (1, 0), (1, 1), (0, 1)])
    b.add_quad([(-5, 5, -5), (-5, 5, 5), (5, 5, 5), (5, 5, -5)], g, toward=(0, 10, 0))
    b.add_point_light((0, 30, 0), (100.0, 100.0, 100.0))
    cam2 = scenes.camera_desc((0, 40, 0), (0, 0, 0), (0, 0, 1), 20.0, 16, 16)
    film2 = scenes.film_desc(16, 16)
    with pytest.raises(ra.EnigmaError) as e:
        r = ra.Renderer(0).configure(b.arrays(), cam2, film2, scenes.direct_lighting_desc(film2, 4, max_depth=3, strategy=abi.DL_SAMPLE_ONE))
        r.render(0, 4)
    assert e.value.code == -4


@pytest.mark.parametrize("fog", [False, True])
def test_image_parity_spheres(fog):
    """Sphere shapes (Enigma/ShapeImp.cpp:13-203, eFloat.h, Transform.h:212-330): full, transformed, cut open, reversed and
    mirrored spheres with all four materials beside the Cornell triangles; with fog, behind a scattering slab."""
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    integs = [scenes.volpath_desc(film, 16)] if fog else [scenes.integrator_desc(film, 16), scenes.direct_lighting_desc(film, 16, max_depth=4, strategy=abi.DL_SAMPLE_ONE),
                                                          scenes.volpath_desc(film, 16)]
    for integ in integs:
        # the direct-lighting integrator's specular rays would carry ray differentials to the textured sphere: untextured there
        sc = scenes.cornell_spheres(fog, textured=not isinstance(integ, abi.EnigmaDirectLightingDesc))
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert np.isfinite(r.image()).all()
        assert common.mean_relative_error(r.image(), o.image()) < 2e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 2e-3 * so[k])
        r.close(); o.close()


def test_image_parity_sphere_area_lights():
    """DiffuseAreaLight on a Sphere (Sphere::sample / pdf, Enigma/ShapeImp.cpp:205-310): the oracle's closed-form scenes --
    cone sampling from outside (also the small-angle series), area sampling from inside a two-sided sphere -- and a
    lamp sphere in the Cornell box with spheres, with and without fog, under the three integrators and both
    light-selection tables that read Power()."""
    from test_oracle_kat import _sphere_light_scene
    cam, film = scenes.camera_desc((0.0, 3.0, -40.0), (0, 2.5, 0), (0, 1, 0), 25.0, 48, 48), scenes.film_desc(48, 48)
    cases = [(_sphere_light_scene(2.0, (0.0, 5.0, 0.0)), cam, film), (_sphere_light_scene(0.1, (0.0, 20.0, 0.0)), cam, film),
             (_sphere_light_scene(50.0, (0.0, 0.0, 0.0), two_sided=True), cam, film)]
    ccam, cfilm = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    cases += [(scenes.cornell_spheres(textured=False, lamp=True), ccam, cfilm), (scenes.cornell_spheres(fog=True, textured=False, lamp=True), ccam, cfilm)]
    for k, (sc, cm, fl) in enumerate(cases):
        integs = [scenes.integrator_desc(fl, 16), scenes.direct_lighting_desc(fl, 16, max_depth=1, strategy=abi.DL_SAMPLE_ALL), scenes.volpath_desc(fl, 16)]
        if k == 3:
            integs += [scenes.integrator_desc(fl, 16, light_strategy=abi.LIGHTS_SPATIAL), scenes.direct_lighting_desc(fl, 16, max_depth=1, strategy=abi.DL_SAMPLE_ONE)]
        if k == 4:
            integs = [scenes.volpath_desc(fl, 16), scenes.volpath_desc(fl, 16, light_strategy=abi.LIGHTS_SPATIAL)]
        for integ in integs:
            r = ra.Renderer(0).configure(sc, cm, fl, integ)
            o = orc.Oracle(sc, cm, fl, integ)
            r.render(0, 16)
            o.render(0, 16, threads=8)
            assert np.isfinite(r.image()).all() and o.image().max() > 1e-6
            assert common.mean_relative_error(r.image(), o.image()) < 2e-3, (k, common.mean_relative_error(r.image(), o.image()))
            so, st = o.stats(), r.stats()
            for key in ("rays_closest", "rays_shadow", "rays_mis"):
                assert abs(st[key] - so[key]) <= max(8, 1e-3 * so[key]), (k, key, st[key], so[key])
            r.close(); o.close()


def test_image_parity_spheres_without_material():
    """A Sphere with Material "none" bounds a medium (Enigma/VolPathIntegrator.cpp:105-111): a scattering ball in front
    of an emissive wall, with and without a lamp sphere above it, under the VolPathIntegrator with both light tables;
    the camera rays, the shadow-ray chains and the MIS rays all cross the ball.  The other integrators refuse."""
    from test_oracle_kat import _fog_ball_scene
    cam, film = scenes.camera_desc((0, 5, -40), (0, 2, 0), (0, 1, 0), 40.0, 48, 48), scenes.film_desc(48, 48)
    for lamp in (False, True):
        sc = _fog_ball_scene((0.02, 0.03, 0.05), (0.15, 0.15, 0.12), g=0.3, lamp=lamp)
        for strategy in (abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL):
            integ = scenes.volpath_desc(film, 32, light_strategy=strategy)
            r = ra.Renderer(0).configure(sc, cam, film, integ)
            o = orc.Oracle(sc, cam, film, integ)
            r.render(0, 32)
            o.render(0, 32, threads=8)
            assert np.isfinite(r.image()).all() and o.image().max() > 0.5
            assert common.mean_relative_error(r.image(), o.image()) < 2e-3, common.mean_relative_error(r.image(), o.image())
            so, st = o.stats(), r.stats()
            for key in ("rays_closest", "rays_shadow", "rays_mis"):
                assert abs(st[key] - so[key]) <= max(8, 1e-3 * so[key]), (key, st[key], so[key])
            r.close(); o.close()
    # DirectLightingIntegrator::Li goes on behind a surface without a material at the same depth (Enigma/DirectLightingIntegrator.cpp:51-54)
    for strat in (abi.DL_SAMPLE_ALL, abi.DL_SAMPLE_ONE):
        integ = scenes.direct_lighting_desc(film, 8, strategy=strat)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 8); o.render(0, 8, threads=8)
        assert o.image().max() > 0
        assert common.mean_relative_error(r.image(), o.image()) < 2e-3
        so, st = o.stats(), r.stats()
        for key in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[key] - so[key]) <= max(8, 1e-3 * so[key]), (key, st[key], so[key])
        r.close(); o.close()


def test_path_integrator_passes_through_surfaces_without_a_material():
    """PathIntegrator::Li skips an intersection without a BSDF: the ray goes on behind it and the vertex is no bounce
    (Enigma/PathIntegrator.cpp:87-93); shadow rays and the BSDF-sampled light rays stop at such surfaces like at any other
    (Scene::intersectP / intersect).  A ball and a box of material-less surfaces in the Cornell box, the ball in front of
    the camera: same image and the same rays as the oracle, and not the image of the scene without them (their shadows)."""
    from test_oracle_kat import _fog_ball_scene
    cam, film = scenes.camera_desc((0, 5, -40), (0, 2, 0), (0, 1, 0), 40.0, 48, 48), scenes.film_desc(48, 48)
    sc = _fog_ball_scene((0.02, 0.03, 0.05), (0.15, 0.15, 0.12), g=0.3, lamp=True)
    for strategy in (abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL):
        integ = scenes.integrator_desc(film, 16, max_depth=3, light_strategy=strategy)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16); o.render(0, 16, threads=8)
        assert o.image().max() > 0.5
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for key in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[key] - so[key]) <= max(4, 1e-4 * so[key]), (key, st[key], so[key])
        assert so["rays_closest"] > 1.2 * so["samples"]          # crossings of the ball are rays of their own
        r.close(); o.close()
    base = scenes.cornell_box("diffuse")
    # triangles without a material: a closed box around the short block of the Cornell box
    v = base.vertices.copy()
    lo, hi = np.array([100.0, 0.5, 100.0]), np.array([300.0, 250.0, 300.0])
    corners = np.array([[lo[0] if i & 1 == 0 else hi[0], lo[1] if i & 2 == 0 else hi[1], lo[2] if i & 4 == 0 else hi[2]] for i in range(8)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = np.array([[q[0], q[1], q[2]] for q in quads] + [[q[0], q[2], q[3]] for q in quads], dtype=np.uint32) + v.shape[0]
    boxed = abi.SceneArrays(np.concatenate([v, corners]), np.concatenate([base.indices, tris]),
                            np.concatenate([base.tri_material, np.full(12, abi.NO_MATERIAL, dtype=np.uint32)]),
                            np.concatenate([base.tri_light, np.full(12, -1, dtype=np.int32)]),
                            list(base.materials)[:base.n_materials], list(base.lights)[:base.n_lights])
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    # the same box under the DirectLightingIntegrator (both strategies, with a smooth glass block whose recursion resumes behind it)
    glassy = abi.SceneArrays(boxed.vertices, boxed.indices, boxed.tri_material, boxed.tri_light,
                             [abi.glass(1.5, 0.0, 0.0, kt=(0, 0, 0)) if i == 0 else m for i, m in enumerate(list(base.materials)[:base.n_materials])],
                             list(base.lights)[:base.n_lights])
    for scene, strat in ((boxed, abi.DL_SAMPLE_ALL), (boxed, abi.DL_SAMPLE_ONE), (glassy, abi.DL_SAMPLE_ALL)):
        integ = scenes.direct_lighting_desc(film, 8, max_depth=4, strategy=strat)
        r = ra.Renderer(0).configure(scene, cam, film, integ)
        o = orc.Oracle(scene, cam, film, integ)
        r.render(0, 8); o.render(0, 8, threads=8)
        assert o.image().max() > 0 and not o.read_past_arrays()
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for key in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[key] - so[key]) <= max(4, 1e-4 * so[key]), (key, st[key], so[key])
        r.close(); o.close()
    integ = scenes.integrator_desc(film, 16)
    imgs = []
    for scene in (boxed, base):
        r = ra.Renderer(0).configure(scene, cam, film, integ)
        r.render(0, 16)
        imgs.append(r.image())
        if scene is boxed:
            o = orc.Oracle(scene, cam, film, integ)
            o.render(0, 16, threads=8)
            assert common.mean_relative_error(r.image(), o.image()) < 1e-3
            so, st = o.stats(), r.stats()
            for key in ("rays_closest", "rays_shadow", "rays_mis"):
                assert abs(st[key] - so[key]) <= max(4, 1e-4 * so[key]), (key, st[key], so[key])
            o.close()
        r.close()
    assert common.mean_relative_error(imgs[0], imgs[1]) > 0.02   # the box is invisible to camera rays but shadows what is inside


def test_spheres_alone_and_refusals():
    """A scene without a single triangle, through the image and the intersection taps; a sphere whose medium index is
    out of range is refused."""
    b = scenes.MeshBuilder()
    b.add_sphere(b.add_material(abi.matte((0.5, 0.5, 0.5))), radius=5.0)
    b.add_sphere(b.add_material(abi.glass(1.5, 0.0, 0.0)), radius=2.0, object_to_world=scenes._translation(0, 9, 0))
    b.add_point_light((10, 30, 5), (3000.0, 3000.0, 3000.0))
    b.add_distant_light((0.2, 1.0, 0.1), (0, 0, 0), (1.0, 1.0, 1.0))
    sc = b.arrays()
    cam, film = scenes.camera_desc((0, 40, 0), (0, 0, 0), (0, 0, 1), 20.0, 48, 48), scenes.film_desc(48, 48)
    integ = scenes.integrator_desc(film, 8, light_strategy=abi.LIGHTS_UNIFORM)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(0, 8); o.render(0, 8, threads=4)
    assert r.image().max() > 0
    assert common.mean_relative_error(r.image(), o.image()) < 1e-3
    rays = np.array([[0, 40, 0, 0, -1, 0, 1e30], [0, 40, 0, 0, 1, 0, 1e30], [0.5, 40, 0.5, 0, -1, 0, 25.0]], np.float32)
    hit = r.intersect(rays)
    ref = o.intersect(rays, ambiguous=False)
    assert hit["tri"].tolist() == [1, -1, -1] == ref["tri"].tolist()      # the glass ball above the matte one; away; tMax short of it
    assert np.array_equal(hit["t"], ref["t"]) and hit["prim_id"].tolist() == [2, 0, 0]
    assert r.intersect_any(rays).tolist() == o.intersect_any(rays).tolist() == [1, 0, 0]
    assert r.stats()["stack_overflows"] == 0
    r.close(); o.close()
    one = scenes.MeshBuilder()
    one.add_sphere(one.add_material(abi.matte((0.5, 0.5, 0.5))), radius=1.0)
    one.add_medium(abi.homogeneous_medium((0.1, 0.1, 0.1), (0.1, 0.1, 0.1), 0.0))
    one.spheres[0].medium_inside = 5                                   # index + 1 beyond the media
    one.add_point_light((0, 5, 0), (10.0, 10.0, 10.0))
    with pytest.raises(ra.EnigmaError) as e:
        ra.Renderer(0).configure(one.arrays(), cam, film, integ)
    assert e.value.code == -1


def _sphere_cloud(n, seed=11, with_floor=True):
    """n small spheres (some partial, some scaled and turned) over a floor of two triangles, lit by a point light and
    an emissive sphere."""
    rng = np.random.default_rng(seed)
    b = scenes.MeshBuilder()
    mats = [b.add_material(abi.matte((0.6, 0.5, 0.4))), b.add_material(abi.glass(1.5, 0.0, 0.0)), b.add_material(abi.matte((0.2, 0.6, 0.7)))]
    if with_floor:
        b.add_quad([(-60, 0, -60), (60, 0, -60), (60, 0, 60), (-60, 0, 60)], mats[0])
    for i in range(n):
        c = rng.uniform((-50, 1, -50), (50, 30, 50))
        m = np.eye(4)
        m[:3, :3] = np.diag(rng.uniform(0.6, 1.6, 3)) if i % 5 == 0 else np.eye(3)
        if i % 7 == 0:
            a = rng.uniform(0, np.pi)
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            m[:3, :3] = rot @ m[:3, :3]
        m[:3, 3] = c
        kw = dict(z_min=-0.5, z_max=0.8, phi_max=270.0) if i % 11 == 0 else {}
        b.add_sphere(mats[i % 3], radius=float(rng.uniform(0.4, 1.5)), object_to_world=m, **kw)
    b.add_point_light((0, 80, 0), (40000.0, 40000.0, 40000.0))
    return b


def test_thousands_of_spheres_in_the_tree():
    """Spheres are leaf entries of the device tree like triangles (BVHAccel holds any Primitive, Enigma/BVH.cpp:148-199):
    5000 of them, full and partial, under rotations and non-uniform scales, through the two intersection taps against
    the oracle's one-by-one tests, and 700 of them through an image."""
    b = _sphere_cloud(5000)
    sc = b.arrays()
    cam, film = scenes.camera_desc((0, 60, -120), (0, 10, 0), (0, 1, 0), 45.0, 64, 64), scenes.film_desc(64, 64)
    integ = scenes.integrator_desc(film, 4, light_strategy=abi.LIGHTS_UNIFORM)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    rng = np.random.default_rng(3)
    n = 6000
    org = rng.uniform((-55, 0.5, -55), (55, 35, 55), (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d, np.where(rng.random(n) < 0.3, rng.uniform(1, 40, n), np.inf)[:, None]], axis=1).astype(np.float32)
    got, ref = r.intersect(rays), o.intersect(rays, ambiguous=False)
    assert (ref["tri"] >= 2).sum() > n // 4 and (ref["tri"] < 0).sum() > 50      # many sphere hits, some misses
    assert np.array_equal(got["tri"], ref["tri"]) and np.array_equal(got["t"], ref["t"])
    assert np.array_equal(got["prim_id"], ref["prim_id"])
    on_tri = (ref["tri"] >= 0) & (ref["tri"] < 2)
    assert np.array_equal(got["bary"][on_tri], ref["bary"][on_tri])
    assert np.array_equal(r.intersect_any(rays), o.intersect_any(rays))
    assert r.stats()["stack_overflows"] == 0
    r.close(); o.close()
    sc = _sphere_cloud(700, seed=5).arrays()
    for make in (scenes.integrator_desc, scenes.volpath_desc):
        integ = make(film, 4, light_strategy=abi.LIGHTS_UNIFORM)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 4); o.render(0, 4, threads=16)
        assert o.image().max() > 0
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for key in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[key] - so[key]) <= max(4, 1e-4 * so[key]), (key, st[key], so[key])
        r.close(); o.close()


def test_image_parity_oren_nayar():
    """Matte with sigma != 0: the OrenNayer lobe (Enigma/BxDF.cpp:744-779) on the Cornell walls and blocks, under the three
    integrators; sigma = 0 and negative sigma stay Lambertian."""
    sc = scenes.cornell_box("diffuse")
    sc.materials[0].sigma = 25.0        # white: walls, blocks
    sc.materials[1].sigma = 60.0        # red
    sc.materials[2].sigma = -5.0        # green: clamped to 0, Lambertian
    cam, film = scenes.cornell_camera(64, 64), scenes.film_desc(64, 64)
    for integ in (scenes.integrator_desc(film, 16), scenes.direct_lighting_desc(film, 16, max_depth=2, strategy=abi.DL_SAMPLE_ALL), scenes.volpath_desc(film, 16)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-4
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 1e-4 * so[k])
        r.close(); o.close()
    lam = scenes.cornell_box("diffuse")
    integ = scenes.integrator_desc(film, 16)
    a, b = ra.Renderer(0).configure(sc, cam, film, integ), ra.Renderer(0).configure(lam, cam, film, integ)
    a.render(0, 16); b.render(0, 16)
    assert common.mean_relative_error(a.image(), b.image()) > 0.02
    a.close(); b.close()


def test_image_parity_spatial_light_distribution():
    """lightsamplestrategy "spatial", the reference's default (Enigma/LightDist.cpp:80-267): one table per voxel from 128 Halton
    points x every light's Sample_Li, made when a path first lands in the voxel (between the closest-hit kernel of a bounce and
    its shading kernels here).  Two lights (Cornell), many lights with the path and the VolPath integrator."""
    cases = [(scenes.cornell_box("microfacet"), "path"), (scenes.many_lights(n_quads=24, n_points=8), "path"), (scenes.cornell_fog("diffuse"), "volpath"),
             (scenes.cornell_box("diffuse", delta_lights=True), "path")]
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    for sc, kind in cases:
        integ = (scenes.volpath_desc if kind == "volpath" else scenes.integrator_desc)(film, 16, light_strategy=abi.LIGHTS_SPATIAL)
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16)
        o.render(0, 16, threads=8)
        assert common.mean_relative_error(r.image(), o.image()) < 1e-3
        so, st = o.stats(), r.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(st[k] - so[k]) <= max(3, 1e-3 * so[k])
        r.close(); o.close()
    # the tables matter: with many lights the spatial pick differs from the power pick sample by sample, yet both are unbiased
    sc = scenes.many_lights(n_quads=24, n_points=8)
    imgs = []
    for strat in (abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL):
        r = ra.Renderer(0).configure(sc, cam, film, scenes.integrator_desc(film, 256, light_strategy=strat))
        r.render(0, 256)
        imgs.append(r.image().astype(np.float64))
        r.close()
    assert not np.allclose(imgs[0], imgs[1], rtol=1e-3)
    assert abs(imgs[0].mean() - imgs[1].mean()) < 0.03 * imgs[0].mean()


def test_spatial_tables_are_made_on_demand_within_a_memory_limit(monkeypatch):
    """Only the voxels that paths reach get a table, so a scene whose table over every voxel would not fit still renders
    (an emitter of many triangles, each a light of its own: Enigma/Api.cpp makes one DiffuseAreaLight per triangle); a limit
    below what the paths need fails the render with a message instead of lighting from a wrong table."""
    sc = scenes.many_lights(n_quads=256, n_points=64)          # 576 lights: 4.6 KB per voxel, 1.2 GB over every voxel
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    integ = scenes.integrator_desc(film, 8, light_strategy=abi.LIGHTS_SPATIAL)
    o = orc.Oracle(sc, cam, film, integ)
    o.render(0, 8, threads=16)
    monkeypatch.setenv("ENIGMA_SPATIAL_TABLE_BYTES", str(128 << 20))  # about 29,000 voxels of 262,144; the paths reach some 17,000
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(0, 4); r.render(4, 8)                              # the second call finds its voxels filled
    assert common.mean_relative_error(r.image(), o.image()) < 1e-3
    r.close()
    monkeypatch.setenv("ENIGMA_SPATIAL_TABLE_BYTES", str(1 << 20))    # 227 voxels: not enough
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 8)
    assert e.value.code == -4 and "voxels" in str(e.value)
    assert not r.film().any()                                   # the failing pass was stopped before its samples reached the film
    r.set_integrator(scenes.integrator_desc(film, 8))           # the power strategy, as the message says
    r.clear()
    r.render(0, 8)
    assert np.isfinite(r.image()).all() and r.image().max() > 0
    r.close(); o.close()


def test_uv_quarter_turn_equals_swapped_roughness_on_the_gpu():
    """The uv axes orient the anisotropic tangent frame: a quarter turn of the uvs equals swapping u / v roughness
    (deterministic with a point light and depth 1)."""
    def floor(ur, vr, axes):
        b = scenes.MeshBuilder()
        m = b.add_material(abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, remap=False, uroughness=ur, vroughness=vr))
        q = [np.array(p, float) for p in ((-50, 0, -50), (-50, 0, 50), (50, 0, 50), (50, 0, -50))]
        for tri in ((0, 1, 2), (0, 2, 3)):
            p = [q[i] for i in tri]
            b.add_tri(p[0], p[1], p[2], m, toward=(0, 1, 0), uvs=tuple((np.dot(axes[0], x) / 100 + .5, np.dot(axes[1], x) / 100 + .5) for x in p))
        b.add_point_light((15, 30, 5), (4000.0, 3000.0, 2000.0))
        return b.arrays()
    ex, ez = np.array([1.0, 0, 0]), np.array([0, 0, 1.0])
    cam, film = scenes.camera_desc((0, 40, 0), (0, 0, 0), (0, 0, 1), 20.0, 32, 32), scenes.film_desc(32, 32)
    integ = scenes.direct_lighting_desc(film, 4, max_depth=1, strategy=abi.DL_SAMPLE_ALL)
    imgs = []
    for ur, vr, axes in ((0.1, 0.5, (ex, ez)), (0.5, 0.1, (ez, -ex)), (0.5, 0.1, (ex, ez))):
        r = ra.Renderer(0).configure(floor(ur, vr, axes), cam, film, integ)
        r.render(0, 4)
        imgs.append(r.image().astype(np.float64))
        r.close()
    assert imgs[0].max() > 0
    assert np.allclose(imgs[0], imgs[1], rtol=2e-4, atol=1e-7)
    assert not np.allclose(imgs[0], imgs[2], rtol=0.05)


@pytest.mark.parametrize("xres,yres", [(37, 23), (8, 61), (67, 8), (5, 3)])
def test_image_parity_sizes_that_are_not_multiples_of_the_pixel_tiles(xres, yres):
    """Pixels are numbered in 8 x 8 tiles with scanline strips right of and below the last full tile: every pixel of
    an odd-sized image must still be rendered exactly once per sample."""
    sc = scenes.cornell_box("diffuse")
    cam, film = scenes.cornell_camera(xres, yres), scenes.film_desc(xres, yres)
    integ = scenes.integrator_desc(film, 4)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(0, 4)
    o.render(0, 4, threads=4)
    assert common.mean_relative_error(r.image(), o.image()) < 1e-4
    assert np.allclose(r.film()[..., 3], o.film_xyzw()[..., 3])          # filter weight sums: one deposit per pixel and sample
    assert r.stats()["camera_samples"] == xres * yres * 4
    r.close(); o.close()


def test_image_parity_mesh_scene():
    sc = scenes.big_mesh(grid=128, sphere_level=4, n_spheres=6)
    film = scenes.film_desc(96, 54)
    integ = scenes.integrator_desc(film, 8)
    cam = scenes.big_mesh_camera(96, 54)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(0, 8)
    o.render(0, 8, threads=8)
    a, b = r.image(), o.image()
    # rough glass / metal paths: a rare libm-level difference can flip a branch; bound the mean, not every pixel
    assert common.mean_relative_error(a, b) < 0.01
    assert np.median(np.abs(a - b) / np.maximum(b, 1e-3)) < 1e-4
    r.close(); o.close()


# ------------------------------------------------------------------ properties / edge cases
def test_sample_ranges_are_additive_and_empty_range_is_a_noop():
    setup = common.cornell_setup(res=64, spp=16, variant="microfacet")
    r = ra.Renderer(0).configure(*setup)
    r.render(0, 16)
    whole = r.film()
    r.clear()
    r.render(0, 5); r.render(5, 5); r.render(5, 16)
    parts = r.film()
    assert np.array_equal(parts[..., 3], whole[..., 3])
    assert np.allclose(parts, whole, rtol=1e-5, atol=1e-5)
    r.close()


def test_pixel_bounds_and_crop_window():
    sc, cam, film, integ = common.cornell_setup(xres=80, yres=60, spp=4, crop=(0.25, 0.75, 0.2, 0.8))
    integ.pixel_bounds[:] = [24, 14, 50, 40]
    r, o = pair((sc, cam, film, integ))
    r.render(0, 4); o.render(0, 4, threads=2)
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])           # exactly the same pixels received samples
    assert (fg[..., 3] > 0).sum() > 0 and (fg[..., 3] == 0).sum() > 0
    assert common.mean_relative_error(r.image(), o.image()) < 1e-4
    r.close(); o.close()


def test_linearity_in_light_radiance():
    sc, cam, film, integ = common.cornell_setup(res=48, spp=8)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(0, 8)
    a = r.image()
    for l in sc.lights[: sc.n_lights]:
        l.L[:] = [v * 2 for v in l.L]
    r2 = ra.Renderer(0).configure(sc, cam, film, integ)
    r2.render(0, 8)
    assert np.allclose(r2.image(), 2 * a, rtol=1e-5, atol=1e-6)   # radiance is linear in Le (light choice pdf unchanged)
    r.close(); r2.close()


def test_full_size_film_invariants_1080p():
    """BASELINE size (1920x1080), one pass: every pixel gets exactly its samples' filter weight, the
    render is repeatable, and two half ranges add up to the whole."""
    sc, cam, film, integ = common.cornell_setup(xres=1920, yres=1080, spp=1024, variant="microfacet")
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    r.render(1020, 1024)
    f = r.film()
    st = r.stats()
    assert st["camera_samples"] == 1920 * 1080 * 4
    # box filter: weight sum = 4 samples per pixel + the double splats of samples on pixel edges
    assert f[..., 3].min() >= 4 and f[..., 3].sum() >= 4 * 1920 * 1080
    extra = f[..., 3].sum() - 4 * 1920 * 1080
    assert extra < 0.01 * 4 * 1920 * 1080
    assert np.isfinite(f).all() and f[..., :3].min() >= 0
    img = ra.resolve(f)
    r.clear(); r.render(1020, 1022); r.render(1022, 1024)
    assert np.allclose(ra.resolve(r.film()), img, rtol=1e-4, atol=1e-5)
    # a 256 x 160 window of the same job (pixel bounds; full-size Sobol' indices, last sample numbers) against the oracle,
    # pixel by pixel (min.x == min.y: the tile-row rule of test_pixel_bounds_and_crop_window)
    integ_w = scenes.integrator_desc(film, 1024)
    integ_w.pixel_bounds[:] = [500, 500, 756, 660]
    r.set_integrator(integ_w)
    r.clear(); r.render(1020, 1024)
    o = orc.Oracle(sc, cam, film, integ_w)
    o.render(1020, 1024, threads=16)
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])
    win = (slice(500, 660), slice(500, 756))
    a, b = r.image()[win], o.image()[win]
    assert np.allclose(img[win][1:-1, 1:-1], a[1:-1, 1:-1], rtol=1e-4, atol=1e-5)    # the window is a piece of the whole job
    assert common.mean_relative_error(a, b) < 1e-4
    # tier 1 on the kernel that renders: the window's primary hits of the first and last sample numbers out of the path state
    _primary_tap_check(r, o, (500, 500, 756, 660), (0, 1023))
    r.close(); o.close()


def test_cornell_512x512_16spp_against_the_oracle():
    """BASELINE configs[0] at its own size: the diffuse Cornell box, 512 x 512, 16 spp, maxdepth 5, the whole image
    against the CPU restatement (same sample numbers, same arithmetic up to libm and summation order).  Tolerances:
    the north star's 1 % mean relative error and 3 sigma per pixel, and the much tighter agreement actually reached."""
    setup = common.cornell_setup(res=512, spp=16, variant="diffuse")
    r, o = pair(setup)
    r.render(0, 16); o.render(0, 16, threads=16)
    a, b = r.image(), o.image()
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])                                     # same pixels, same filter weights
    mre = common.mean_relative_error(a, b)
    assert mre < 0.01
    assert mre < 1e-4
    # 3 sigma of the Monte Carlo estimator per pixel: sigma from the spread of 16 one-sample images would need 16 more
    # renders; the two images use the SAME samples, so their difference is far below any sampling noise: bound it directly
    assert np.abs(a - b).max() < 0.05 * max(1.0, float(b.max()))
    gs, os_ = r.stats(), o.stats()
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(gs[k]) - int(os_[k])) <= max(2, 1e-5 * os_[k]), k               # the same paths, ray for ray (exact-distance ties aside)
    r.close(); o.close()


def test_many_lights_512x512_16spp_against_the_oracle():
    """BASELINE configs[3] at its own size: 512 emissive triangles + 64 point lights, 512 x 512, 16 spp, the whole image
    against the CPU restatement, for the two integrators the configuration compares (PathIntegrator with the power
    distribution, DirectLightingIntegrator sampling one light) and for the reference's default light strategy (spatial)."""
    sc = scenes.many_lights()
    cam, film = scenes.cornell_camera(512, 512), scenes.film_desc(512, 512)
    for integ in (scenes.integrator_desc(film, 16), scenes.direct_lighting_desc(film, 16, strategy=abi.DL_SAMPLE_ONE),
                  scenes.integrator_desc(film, 16, light_strategy=abi.LIGHTS_SPATIAL)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 16); o.render(0, 16, threads=16)
        a, b = r.image(), o.image()
        assert np.array_equal(r.film()[..., 3], o.film_xyzw()[..., 3])
        mre = common.mean_relative_error(a, b)
        assert mre < 0.01 and mre < 1e-4, mre
        gs, os_ = r.stats(), o.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):   # 576 lights: now and then the last bit of a cosine decides whether a sample faces its light
            assert abs(int(gs[k]) - int(os_[k])) <= max(2, 5e-5 * os_[k]), (k, gs[k], os_[k])
        r.close(); o.close()


def test_full_size_big_mesh_1080p():
    """The >= 1 M triangle configuration at its full size (1,179,662 triangles, 1920x1080 sampler).  The whole
    film is checked through invariants; a 192x108 window of the same job (pixel bounds, full-size Sobol indices,
    full scene) is compared with the oracle pixel by pixel."""
    sc = scenes.big_mesh()
    assert sc.indices.shape[0] >= 1 << 20
    cam = scenes.big_mesh_camera(1920, 1080)
    film = scenes.film_desc(1920, 1080)
    integ = scenes.integrator_desc(film, 1024)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    assert r.stats()["bvh_nodes"] > 100000
    r.render(0, 2)
    f = r.film()
    st = r.stats()
    assert st["camera_samples"] == 1920 * 1080 * 2
    assert f[..., 3].min() >= 2 and f[..., 3].sum() - 2 * 1920 * 1080 < 0.01 * 2 * 1920 * 1080
    assert np.isfinite(f).all() and f[..., :3].min() >= 0
    assert (f[..., :3].sum(axis=2) > 0).mean() > 0.9            # closed sky box + overhead light: almost no black pixels
    whole = ra.resolve(f)
    r.clear(); r.render(0, 1); r.render(1, 2)
    assert np.allclose(ra.resolve(r.film()), whole, rtol=1e-4, atol=1e-5)
    # window of the same job against the oracle (min.x == min.y: see the tile-row rule in test_pixel_bounds_and_crop_window)
    integ_w = scenes.integrator_desc(film, 1024)
    integ_w.pixel_bounds[:] = [600, 600, 792, 708]
    r.set_integrator(integ_w)
    r.clear(); r.render(0, 2)
    o = orc.Oracle(sc, cam, film, integ_w)
    o.render(0, 2, threads=16)
    fg, fo = r.film(), o.film_xyzw()
    assert np.array_equal(fg[..., 3], fo[..., 3])
    win = (slice(600, 708), slice(600, 792))
    a, b = r.image()[win], o.image()[win]
    assert np.allclose(whole[win][1:-1, 1:-1], a[1:-1, 1:-1], rtol=1e-4, atol=1e-5)   # the window is a piece of the whole job
    assert common.mean_relative_error(a, b) < 0.01
    assert np.median(np.abs(a - b) / np.maximum(b, 1e-3)) < 1e-5
    # tier 1 on the kernel that renders (the HBM build, 1.18 M triangles): the window's primary hits out of the path state
    n, hits = _primary_tap_check(r, o, (600, 600, 792, 708), (0, 1023))
    assert hits == n                                            # closed sky box: every camera ray hits something
    r.close(); o.close()


def test_bsdf_point_evaluations_match_the_golden_fixture():
    """BSDF::f, Pdf and sample_f (Enigma/BxDF.cpp:885-985) of the five material kinds at sixteen (wo, wi, u) triples each,
    device against the fixture the oracle wrote (tools/make_golden.py) and against the oracle itself.  Values agree to a few
    ulp (libm: sinf, cosf, tanf, atanf, sqrtf orders), sampled lobe types exactly."""
    g = np.load(os.path.join(GOLD, "bsdf_points.npz"))
    mats = {"matte": abi.matte((0.6, 0.5, 0.4)), "metal": abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, 0.1),
            "glass_rough": abi.glass(1.5, 0.05, 0.08), "glass_smooth": abi.glass(1.5, 0, 0), "plastic": abi.plastic((0.3, 0.3, 0.6), (0.4, 0.4, 0.4), 0.05)}
    r = ra.Renderer(0)
    wo, wi, u = g["wo"], g["wi"], g["u"]
    for name, m in mats.items():
        got, want = r.bsdf_eval(m, wo, wi, u), g[name]
        assert np.array_equal(orc.bsdf_eval(m, wo, wi, u), want), name                 # the fixture is what the oracle says today
        assert np.array_equal(got[:, 11], want[:, 11]), name                            # same lobe sampled
        scale = np.maximum(np.abs(want[:, :11]), 1e-6)
        assert (np.abs(got[:, :11] - want[:, :11]) / scale).max() < 2e-4, (name, (np.abs(got[:, :11] - want[:, :11]) / scale).max())
    # a rough matte (Oren-Nayar) and anisotropic metal against the oracle directly
    for m in (abi.matte((0.5, 0.6, 0.7), sigma=30.0), abi.metal(scenes.COPPER_ETA, scenes.COPPER_K, uroughness=0.05, vroughness=0.3)):
        got, want = r.bsdf_eval(m, wo, wi, u), orc.bsdf_eval(m, wo, wi, u)
        assert np.array_equal(got[:, 11], want[:, 11])
        scale = np.maximum(np.abs(want[:, :11]), 1e-6)
        assert (np.abs(got[:, :11] - want[:, :11]) / scale).max() < 2e-4
    r.close()


def test_error_conventions():
    r = ra.Renderer(0)
    with pytest.raises(ra.EnigmaError) as e:
        r.render(0, 1)
    assert e.value.code == -2                                  # ENIGMA_E_STATE: nothing configured
    sc, cam, film, integ = common.cornell_setup(res=16, spp=4)
    bad = scenes.cornell_box("diffuse")
    bad.materials[0].type = 17
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(bad)
    assert e.value.code == -4                                  # ENIGMA_E_UNSUPPORTED: a material the path does not cover
    bad2 = scenes.cornell_box("diffuse")
    bad2.indices[3, 1] = 10 ** 6
    with pytest.raises(ra.EnigmaError) as e:
        r.upload_scene(bad2)
    assert e.value.code == -1                                  # ENIGMA_E_INVALID
    r.configure(sc, cam, film, integ)
    with pytest.raises(ra.EnigmaError):
        r.render(0, 99)                                        # beyond samples_per_pixel
    integ2 = scenes.integrator_desc(film, 4)
    integ2.light_strategy = 7
    with pytest.raises(ra.EnigmaError) as e:
        r.set_integrator(integ2)
    assert e.value.code == -4                                  # no such light distribution
    r.render(0, 4)                                             # still usable after errors
    assert np.isfinite(r.image()).all()
    r.close()


def test_scene_without_lights_and_empty_scene():
    b = scenes.MeshBuilder()
    m = b.add_material(abi.matte((0.5, 0.5, 0.5)))
    b.add_quad([(0, 0, 500), (556, 0, 500), (556, 548, 500), (0, 548, 500)], m, toward=(278, 273, 0))
    sc = b.arrays()
    film = scenes.film_desc(32, 32)
    integ = scenes.integrator_desc(film, 4)
    cam = scenes.cornell_camera(32, 32)
    r = ra.Renderer(0).configure(sc, cam, film, integ)
    o = orc.Oracle(sc, cam, film, integ)
    r.render(0, 4); o.render(0, 4, threads=1)
    assert np.array_equal(r.image(), np.zeros_like(r.image())) and np.array_equal(o.image(), r.image())
    assert r.stats()["rays_shadow"] == 0
    r.close(); o.close()
    empty = abi.SceneArrays(np.zeros((0, 3)), np.zeros((0, 3)), [], [], [abi.matte((0.5, 0.5, 0.5))], [])
    r = ra.Renderer(0).configure(empty, cam, film, integ)
    r.render(0, 4)
    assert r.film()[..., 3].min() >= 4 and r.image().max() == 0
    r.close()


def _nested_shelves(n=120, ratio=0.82):
    """A small scene whose tree is deep: quads in a geometric progression of sizes towards a corner (each split of the
    builder peels the few largest off the rest), above a floor, under a lamp.  A few hundred triangles: it runs from LDS."""
    b = scenes.MeshBuilder()
    white = b.add_material(abi.matte((0.7, 0.7, 0.7)))
    red = b.add_material(abi.matte((0.7, 0.2, 0.2)))
    metal = b.add_material(abi.metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), roughness=0.05))
    b.add_quad([(-300, 0, -300), (-300, 0, 300), (300, 0, 300), (300, 0, -300)], white, toward=(0, 1, 0))
    x, s = -200.0, 80.0
    for k in range(n):
        y = 20.0 + 3.0 * (k % 5)
        m = (white, red, metal)[k % 3]
        b.add_quad([(x, y, -s), (x, y, s), (x + s, y + 0.3 * s, s), (x + s, y + 0.3 * s, -s)], m, toward=(x, 1e4, 0))
        x += s * 1.05
        s *= ratio
    b.add_quad([(-60, 250, -60), (-60, 250, 60), (60, 250, 60), (60, 250, -60)], white, toward=(0, -1, 0), emit=(25.0, 25.0, 25.0))
    return b.arrays()


def test_small_scene_with_a_deep_tree_takes_the_deep_lds_build():
    """Scenes that fit LDS run one of two builds of the traversal kernels: trees up to six levels deep (every Cornell
    scene) keep six stack levels per lane, deeper ones twelve.  This scene is small AND deep: hits bit for bit, image and
    ray counts against the oracle, for the three integrators' builds of the kernel."""
    sc = _nested_shelves()
    cam = scenes.camera_desc((150, 260, -420), (120, 10, 0), (0, 1, 0), 40.0, 96, 64)
    film = scenes.film_desc(96, 64)
    for integ in (scenes.integrator_desc(film, 16), scenes.direct_lighting_desc(film, 16, max_depth=3, strategy=abi.DL_SAMPLE_ONE), scenes.volpath_desc(film, 16)):
        r = ra.Renderer(0).configure(sc, cam, film, integ)
        o = orc.Oracle(sc, cam, film, integ)
        r.render(0, 8); o.render(0, 8, threads=8)
        st = r.stats()
        assert st["scene_in_lds"] == 1 and 6 < st["bvh_depth"] <= 12, (st["scene_in_lds"], st["bvh_depth"], st["bvh_nodes"])
        assert common.mean_relative_error(r.image(), o.image()) < 1e-4
        so = o.stats()
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert st[k] == so[k], (k, st[k], so[k])
        _primary_tap_check(r, o, (0, 0, 96, 64), (0, 5))
        r.close(); o.close()
