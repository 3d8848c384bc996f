#!/usr/bin/env python3
"""Differential fuzzing of the GPU library against the CPU oracle on random small scenes (needs a GPU).

    python tools/fuzz_scenes.py <first seed> <count> [out.json]

A seed fixes everything: a soup of random triangles (some degenerate, some tiny, some huge, some with per-vertex
normals / uvs), random materials of every kind (the lobe-list ones -- mirror, translucent, substrate, uber, mix -- through a stream of their own), emissive triangles, delta lights, optionally spheres, media and an
environment light, a random camera, integrator (path / direct lighting / volpath), depth, Russian-roulette threshold,
light strategy and pixel filter; a second stream of choices adds image textures, a dense bumpy sheet (a deep tree),
surfaces without a material and a crop window.  Per seed the script compares ray counts and images and prints one
line; failures (an error code on one side only, other than the documented refusals; counts apart by more than 1 %;
mean relative error above 1 %) end in a non-zero exit status.  tests/test_gpu_random_scenes.py runs a fixed handful of seeds through random_case().
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytrace_amd import abi, scenes  # noqa: E402


DOCUMENTED_REFUSALS = ("reads past the sample arrays", "vertex normals lie on both sides", "trilinear textures", "area light on a surface without a material",
                       "SpecularTransmit picks its lobe", "two trilinear textures")


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def _colour(rng, lo=0.05, hi=0.95):
    return tuple(float(x) for x in rng.uniform(lo, hi, 3))


def random_material(rng):
    k = rng.randint(0, 7)
    if k == 0:
        return abi.matte(_colour(rng))
    if k == 1:
        return abi.matte(_colour(rng), sigma=float(rng.uniform(5, 80)))
    if k == 2:
        if rng.rand() < 0.5:
            return abi.metal(_colour(rng, 0.1, 2.0), _colour(rng, 1.0, 5.0), roughness=float(10 ** rng.uniform(-3, -0.3)), remap=bool(rng.randint(2)))
        return abi.metal(_colour(rng, 0.1, 2.0), _colour(rng, 1.0, 5.0), uroughness=float(10 ** rng.uniform(-2.5, -0.5)), vroughness=float(10 ** rng.uniform(-2.5, -0.5)), remap=bool(rng.randint(2)))
    if k == 3:
        return abi.glass(float(rng.uniform(1.05, 2.2)), 0.0, 0.0, kr=_colour(rng), kt=_colour(rng))
    if k == 4:
        return abi.glass(float(rng.uniform(1.05, 2.2)), float(10 ** rng.uniform(-2.5, -0.5)), float(10 ** rng.uniform(-2.5, -0.5)), kr=_colour(rng), kt=_colour(rng), remap=bool(rng.randint(2)))
    if k == 5:
        return abi.plastic(_colour(rng), _colour(rng), roughness=float(10 ** rng.uniform(-2.5, -0.3)), remap=bool(rng.randint(2)))
    return abi.glass(1.5, 0.0, 0.0, kr=(1, 1, 1), kt=(0, 0, 0))  # a mirror-like glass: reflection only


def random_lobe_list_material(rng, materials, plain):
    """A mirror, translucent, substrate or uber material, or a mix over two of `plain` (indices of materials without
    textures) -- now and then with black colours, full or no opacity, weights beyond 1."""
    c = lambda: _colour(rng) if rng.rand() < 0.85 else ((0.0, 0.0, 0.0) if rng.rand() < 0.5 else (float(rng.rand() < 0.5), 0.0, float(rng.rand() < 0.5)))
    rough = lambda: float(10 ** rng.uniform(-2.5, -0.3))
    k = rng.randint(0, 6 if len(plain) else 4)
    if k == 0:
        return abi.mirror(c())
    if k == 1:
        return abi.translucent(c(), c(), c(), c(), roughness=rough(), remap=bool(rng.randint(2)))
    if k == 2:
        return abi.substrate(c(), c(), rough(), rough(), remap=bool(rng.randint(2)))
    if k == 3:
        op = float(rng.choice([1.0, 1.0, 0.0, rng.uniform(0.1, 0.9)]))
        opacity = (op, op, op) if rng.rand() < 0.7 else _colour(rng, 0.0, 1.0)
        kr, kt = (c() if rng.rand() < 0.5 else (0.0, 0.0, 0.0)), (c() if rng.rand() < 0.5 else (0.0, 0.0, 0.0))
        return abi.uber(c(), c(), kr, kt, opacity, uroughness=rough(), vroughness=rough(), index=float(rng.uniform(1.0, 2.2)), remap=bool(rng.randint(2)))
    a, b_ = int(plain[rng.randint(len(plain))]), int(plain[rng.randint(len(plain))])
    amount = _colour(rng, 0.0, 1.0) if rng.rand() < 0.8 else tuple(float(x) for x in rng.choice([0.0, 1.0, 1.5, -0.5], 3))
    return abi.mix(a, b_, amount)


def random_case(seed, res=24, spp=4, for_file=False, details=None):
    """(scene arrays, camera, film, integrator description, label) of one seed.
    for_file: the variant tests/pbrt_writer.py can write down (lights on spheres last, a constant environment);
    details: a dictionary that receives what the scene file's header needs (camera, filter, integrator, crop)."""
    rng = np.random.RandomState(seed)
    b = scenes.MeshBuilder()
    mats = [b.add_material(random_material(rng)) for _ in range(rng.randint(1, 7))]
    extent = 10.0
    integ_kind = rng.choice(["path", "path", "path", "direct-one", "direct-all", "volpath"])
    use_media = integ_kind == "volpath" and rng.rand() < 0.8
    med = -1
    if use_media:
        med = b.add_medium(abi.homogeneous_medium(_colour(rng, 0.0, 0.05), _colour(rng, 0.0, 0.15), g=float(rng.uniform(-0.7, 0.7))))
    # a closed room (sometimes), so that most paths keep bouncing
    if rng.rand() < 0.6:
        lo, hi = -2.0, extent + 2.0
        c = [(lo, lo, lo), (hi, lo, lo), (hi, hi, lo), (lo, hi, lo), (lo, lo, hi), (hi, lo, hi), (hi, hi, hi), (lo, hi, hi)]
        centre = ((lo + hi) / 2,) * 3
        for f in ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 3, 7, 4)):
            b.add_quad([c[i] for i in f], mats[rng.randint(len(mats))], toward=centre)
    n_tris = rng.randint(4, 60)
    n_emit = 0
    for i in range(n_tris):
        centre = rng.uniform(0, extent, 3)
        size = 10 ** rng.uniform(-1.0, 0.9)
        p = [centre + rng.normal(size=3) * size for _ in range(3)]
        kind = rng.rand()
        if kind < 0.04:
            p[2] = p[1].copy()                      # two equal vertices
        elif kind < 0.08:
            p[2] = p[0] + 2.5 * (p[1] - p[0])       # collinear
        elif kind < 0.12:
            p = [centre + rng.normal(size=3) * 1e-4 for _ in range(3)]  # tiny
        mat = mats[rng.randint(len(mats))]
        emit = None
        if rng.rand() < 0.12 or (i == n_tris - 1 and n_emit == 0 and rng.rand() < 0.7):
            emit = tuple(float(x) for x in rng.uniform(2, 30, 3))
            n_emit += 1
        normals = uvs = None
        if rng.rand() < 0.25:
            g = np.cross(p[0] - p[2], p[1] - p[2])
            gl = np.linalg.norm(g)
            g = g / gl if gl > 0 else np.array([0.0, 0.0, 1.0])
            if rng.rand() < 0.5:
                g = -g
            normals = [g + rng.normal(size=3) * 0.3 for _ in range(3)]
        if rng.rand() < 0.25:
            uvs = [rng.uniform(0, 1, 2) for _ in range(3)]
            if rng.rand() < 0.15:
                uvs[2] = uvs[1]                     # degenerate uvs
        if use_media and rng.rand() < 0.2:
            b.medium_interface(med if rng.rand() < 0.5 else -1, med if rng.rand() < 0.7 else -1)
        else:
            b.medium_interface(med if use_media and rng.rand() < 0.3 else -1, med if use_media and rng.rand() < 0.3 else -1) if use_media else b.medium_interface(-1, -1)
        b.add_tri(p[0], p[1], p[2], mat, emit=emit, two_sided=bool(rng.randint(2)), normals=normals, uvs=uvs,
                  reverse_orientation=bool(rng.rand() < 0.2))
    b.medium_interface(-1, -1)
    def add_spheres():
        for _ in range(rng.randint(1, 4)):
                m = np.eye(4)
                m[:3, 3] = rng.uniform(0, extent, 3)
                if rng.rand() < 0.5:
                    m[:3, :3] = scenes._rotation(_unit(rng), float(rng.uniform(0, 360)))[:3, :3]
                kw = {}
                if rng.rand() < 0.3:
                    kw = dict(z_min=float(-rng.uniform(0.2, 1.0)), z_max=float(rng.uniform(0.2, 1.0)), phi_max=float(rng.uniform(90, 360)))
                emit = tuple(float(x) for x in rng.uniform(2, 20, 3)) if rng.rand() < 0.3 else None
                r = float(rng.uniform(0.3, 2.0))
                if kw:
                    kw["z_min"] *= r; kw["z_max"] *= r
                b.add_sphere(mats[rng.randint(len(mats))], r, m, emit=emit, two_sided=bool(rng.randint(2)), reverse_orientation=bool(rng.rand() < 0.2), **kw)
    want_spheres = integ_kind != "direct-all" and rng.rand() < 0.3
    if want_spheres and not for_file:
        add_spheres()
    if rng.rand() < 0.35:
        b.add_point_light(tuple(rng.uniform(0, extent, 3)), tuple(float(x) for x in rng.uniform(20, 200, 3)))
    if rng.rand() < 0.25:
        b.add_spot_light(tuple(rng.uniform(0, extent, 3)), tuple(rng.uniform(0, extent, 3)), tuple(float(x) for x in rng.uniform(50, 400, 3)),
                         coneangle=float(rng.uniform(10, 80)), conedeltaangle=float(rng.uniform(1, 9)))
    if rng.rand() < 0.2:
        b.add_distant_light(tuple(rng.uniform(-1, 1, 3) * 2), (0, 0, 0), tuple(float(x) for x in rng.uniform(0.5, 3, 3)))
    if rng.rand() < 0.2:
        b.lights.append(abi.infinite_light(tuple(float(x) for x in rng.uniform(0.2, 1.5, 3)), radiance_map=scenes.sky_map() if (rng.rand() < 0.5 and (not for_file or seed >= 1800000)) else None))   # scene files carry the map as a PFM file (seeds from 1,800,000 on)
    # an eighth stream, for seeds from 1,800,000 on: a projection light (a point light whose weight in the light distributions is its cone's)
    pr = np.random.RandomState((seed * 7919 + 0x9207) & 0x7fffffff)
    if seed >= 1800000 and pr.rand() < 0.3:
        b.lights.append(abi.projection_light(tuple(pr.uniform(0, extent, 3)), tuple(float(x) for x in pr.uniform(20, 300, 3)), fov=float(pr.uniform(5, 150))))
    if seed >= 1900000 and pr.rand() < 0.3:   # and from 1,900,000 on a goniometric light, with a map of any size in two cases of three
        gmap = scenes.sky_map(int(pr.randint(1, 20)), int(pr.randint(1, 40))) if pr.rand() < 0.67 else None
        b.lights.append(abi.goniometric_light(tuple(pr.uniform(0, extent, 3)), tuple(float(x) for x in pr.uniform(20, 300, 3)), gmap))
    if want_spheres and for_file:
        add_spheres()
    # ---- a second stream of choices (textures, a dense piece of mesh, boundary surfaces, a crop window): drawn apart from
    # the first so that a seed's basic scene stays what it was when a choice is added here
    xr = np.random.RandomState((seed * 2654435761 + 0x5EED) & 0x7fffffff)
    if xr.rand() < 0.3:
        for _ in range(xr.randint(1, 4)):
            img = scenes.test_pattern(int(xr.choice([2, 5, 16, 32])), int(xr.choice([2, 7, 16, 64])), seed=int(xr.randint(1 << 30)))
            t = b.add_texture(abi.image_texture(img * float(xr.uniform(0.3, 1.0)), trilinear=bool(xr.rand() < 0.7), gamma=bool(xr.rand() < 0.8),
                                                scale=float(xr.uniform(0.5, 1.0)), wrap=int(xr.randint(3)), su=float(xr.uniform(0.3, 4)), sv=float(xr.uniform(0.3, 4)),
                                                du=float(xr.uniform(-1, 1)), dv=float(xr.uniform(-1, 1))))
            m = b.materials[xr.randint(len(b.materials))]
            field = {abi.MAT_MATTE: ["tex_kd"], abi.MAT_METAL: ["tex_eta", "tex_k"], abi.MAT_GLASS: ["tex_kr", "tex_kt"], abi.MAT_PLASTIC: ["tex_kd", "tex_ks"]}[m.type]
            setattr(m, field[xr.randint(len(field))], t)
    if xr.rand() < 0.25:  # a bumpy sheet of many small triangles: a deep tree under a shallow one
        n = int(xr.choice([8, 24, 48]))
        org, ax, ay = xr.uniform(1, 6, 3), _unit(xr) * xr.uniform(2, 6), _unit(xr) * xr.uniform(2, 6)
        up = np.cross(ax, ay); up = up / max(np.linalg.norm(up), 1e-9)
        hgt = xr.uniform(-0.15, 0.15, (n + 1, n + 1))
        P = lambda i, j: org + ax * (i / n) + ay * (j / n) + up * hgt[i, j]
        mat = mats[xr.randint(len(mats))]
        smooth = xr.rand() < 0.5
        for i in range(n):
            for j in range(n):
                q = [P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)]
                uv = [(i / n, j / n), ((i + 1) / n, j / n), ((i + 1) / n, (j + 1) / n), (i / n, (j + 1) / n)]
                nrm = [up + xr.normal(size=3) * 0.05 for _ in range(4)] if smooth else None
                b.add_tri(q[0], q[1], q[2], mat, uvs=(uv[0], uv[1], uv[2]), normals=None if nrm is None else (nrm[0], nrm[1], nrm[2]))
                b.add_tri(q[0], q[2], q[3], mat, uvs=(uv[0], uv[2], uv[3]), normals=None if nrm is None else (nrm[0], nrm[2], nrm[3]))
    if xr.rand() < 0.2:  # surfaces without a material: passed through by all three integrators
        for _ in range(xr.randint(1, 4)):
            centre = xr.uniform(0, extent, 3)
            if use_media:
                b.medium_interface(med if xr.rand() < 0.7 else -1, med if xr.rand() < 0.3 else -1)
            p = [centre + xr.normal(size=3) * 3 for _ in range(3)]
            b.add_tri(p[0], p[1], p[2], abi.NO_MATERIAL)
        b.medium_interface(-1, -1)
    # ---- a third stream: parameters at the edges of their ranges (black colours, index 1, zero roughness, media without
    # scattering or absorption, |g| near 1, black lights, a light sitting on a vertex, textures one texel wide)
    er = np.random.RandomState((seed * 40503 + 0xED6E) & 0x7fffffff)
    if er.rand() < 0.35:
        for m in b.materials:
            k = er.rand()
            if k < 0.15:
                for f in ("kd", "ks", "kr", "kt"):
                    getattr(m, f)[:] = [0.0, 0.0, 0.0] if er.rand() < 0.5 else [float(er.rand() < 0.5), 0.0, float(er.rand() < 0.5)]
            elif k < 0.3 and (m.type != abi.MAT_GLASS or (m.uroughness == 0 and m.vroughness == 0)):
                m.index = 1.0  # (a rough glass of index 1 is left out: its half vector wo + wi is rounding noise, in the reference too)
            elif k < 0.45:
                m.uroughness = m.vroughness = 0.0
            elif k < 0.55:
                m.k[:] = [0.0, 0.0, 0.0]; m.eta[:] = [1.0, 1.0, 1.0]
            elif k < 0.6:
                m.uroughness, m.vroughness = 1.0, float(er.uniform(1.0, 3.0))
        for md in b.media:
            k = er.rand()
            if k < 0.2: md.sigma_s[:] = [0.0, 0.0, 0.0]
            elif k < 0.4: md.sigma_a[:] = [0.0, 0.0, 0.0]
            elif k < 0.5: md.sigma_a[:] = [0.0, 0.0, 0.0]; md.sigma_s[:] = [0.0, 0.0, 0.0]
            elif k < 0.6: md.sigma_s[:] = [0.0, float(er.uniform(0, 0.3)), 0.0]
            if er.rand() < 0.3: md.g = float(er.choice([-0.999, 0.999, -1.0, 1.0]))
        for l in b.lights:
            k = er.rand()
            if k < 0.15: l.L[:] = [0.0, 0.0, 0.0]
            elif k < 0.25: l.L[:] = [0.0, float(l.L[1]), 0.0]
            elif k < 0.35 and l.type in (abi.LIGHT_POINT, abi.LIGHT_SPOT) and b.v:
                l.pos[:] = [float(x) for x in np.float32(b.v[er.randint(len(b.v))])]
        if er.rand() < 0.1:
            for l in b.lights: l.L[:] = [0.0, 0.0, 0.0]
    if er.rand() < 0.15 and b.materials:
        h_, w_ = int(er.choice([1, 1, 3, 17])), int(er.choice([1, 2, 5, 33]))
        t = b.add_texture(abi.image_texture(er.uniform(0, 1, (h_, w_, 3)).astype(np.float32), trilinear=bool(er.rand() < 0.7), wrap=int(er.randint(3)),
                                            su=float(er.uniform(0.3, 40)), sv=float(er.uniform(0.3, 40))))
        m = b.materials[er.randint(len(b.materials))]
        field = {abi.MAT_MATTE: ["tex_kd"], abi.MAT_METAL: ["tex_eta", "tex_k"], abi.MAT_GLASS: ["tex_kr", "tex_kt"], abi.MAT_PLASTIC: ["tex_kd", "tex_ks"]}[m.type]
        setattr(m, field[er.randint(len(field))], t)
    # ---- a fifth stream: materials that are lists of lobes (mirror, translucent, substrate, uber, mix) take the place of
    # some of the scene's materials; a mix names materials made before it, itself a mix now and then
    gr = np.random.RandomState((seed * 22695477 + 0x10BE) & 0x7fffffff)
    if gr.rand() < 0.3:
        moved = {}
        for i in range(len(b.materials)):
            if gr.rand() < 0.5:
                new = random_lobe_list_material(gr, b.materials, list(range(len(b.materials))))   # a mix may name textured materials
                if b.textures and new.type != abi.MAT_MIX and gr.rand() < 0.4:
                    fields = {abi.MAT_MIRROR: ["tex_kr"], abi.MAT_TRANSLUCENT: ["tex_kd", "tex_ks"], abi.MAT_SUBSTRATE: ["tex_kd", "tex_ks"],
                              abi.MAT_UBER: ["tex_kd", "tex_ks", "tex_kr", "tex_kt"]}[new.type]
                    for f in fields:
                        if gr.rand() < 0.5:
                            setattr(new, f, int(gr.randint(len(b.textures))) + 1)
                b.materials.append(new)
                moved[i] = len(b.materials) - 1
        b.mat = [moved.get(m, m) for m in b.mat]
        for sp in b.spheres:
            if sp.material in moved:
                sp.material = moved[sp.material]
    crop = (0.0, 1.0, 0.0, 1.0)
    if xr.rand() < 0.2:
        x0, y0 = xr.uniform(0, 0.5, 2)
        crop = (float(x0), float(x0 + xr.uniform(0.2, 0.5)), float(y0), float(y0 + xr.uniform(0.2, 0.5)))
    sc = b.arrays(weld=bool(rng.randint(2)))
    eye = rng.uniform(0, extent, 3)
    look = rng.uniform(2, extent - 2, 3)
    if np.linalg.norm(look - eye) < 1e-2:
        look = eye + np.array([0, 0, 1.0])
    up = _unit(rng)
    if abs(np.dot(up, (look - eye) / np.linalg.norm(look - eye))) > 0.95:
        up = np.cross(up, look - eye); up /= np.linalg.norm(up)
    lens = float(rng.uniform(0.01, 0.2)) if rng.rand() < 0.2 else 0.0
    f32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)  # what a scene file can carry: single-precision parameters
    if for_file:
        eye, look, up, lens = f32(eye), f32(look), f32(up), float(f32(lens))
    xres, yres = res + rng.randint(0, 9), res + rng.randint(0, 9)
    fov, focal = float(rng.uniform(25, 100)), float(rng.uniform(2, 12))
    if for_file:
        fov, focal = float(f32(fov)), float(f32(focal))
        crop = tuple(float(f32(c)) for c in crop)
    cam = scenes.camera_desc(tuple(eye), tuple(look), tuple(up), fov, xres, yres, lens_radius=lens, focal_distance=focal if lens else 1e30)
    # ---- a sixth stream: the environment camera now and then, and (scene files) the meshes in PLY files
    cr = np.random.RandomState((seed * 1103515245 + 0xCA40) & 0x7fffffff)
    environment = bool(cr.rand() < 0.12)
    ply = bool(cr.rand() < 0.3)
    if environment:
        cam = scenes.environment_camera_desc(tuple(eye), tuple(look), tuple(up))
    fk = rng.choice(["box", "box", "triangle", "gaussian", "mitchell", "sinc"])
    # ---- a fourth stream: film and integrator options (filter radii, a luminance clamp, pixel bounds, samples per light,
    # more samples per pixel rendered in several calls)
    orr = np.random.RandomState((seed * 69069 + 0x0F17) & 0x7fffffff)
    film_kw = {}
    pixel_bounds = None
    splits = None
    if not for_file and orr.rand() < 0.4:
        if orr.rand() < 0.4:
            film_kw["radius"] = (float(orr.uniform(0.5, 3.5)), float(orr.uniform(0.5, 3.5)))
        if orr.rand() < 0.25:
            film_kw["max_sample_luminance"] = float(10 ** orr.uniform(-1, 2))
        if orr.rand() < 0.3:
            x0, y0 = int(orr.randint(0, xres // 2)), int(orr.randint(0, yres // 2))
            x0 = max(x0, y0)   # rows start at min.x in the reference's tile loop: min.x < min.y is undefined there
            pixel_bounds = [x0, y0, int(orr.randint(x0 + 1, xres + 1)), int(orr.randint(y0 + 1, yres + 1))]
        if orr.rand() < 0.3:
            for l in b.lights:
                l.n_samples = int(orr.choice([1, 2, 3, 4, 8]))
            sc = b.arrays(weld=bool(orr.randint(2)))
        if orr.rand() < 0.3:
            spp = int(orr.choice([1, 2, 8, 16]))
            cuts = sorted(set(int(c) for c in orr.randint(0, spp + 1, size=2)))
            splits = [0] + cuts + [spp]
    # ---- a seventh stream, for seeds from 1,700,000 on (the seeds recorded before keep their scenes): the HaltonSampler in
    # place of the Sobol' one, with sample counts that are no powers of two, now and then all samples at the pixel centres
    hr = np.random.RandomState((seed * 134775813 + 0x4A17) & 0x7fffffff)
    sampler, center = "sobol", False
    if seed >= 1700000 and hr.rand() < 0.35:
        sampler, center = "halton", bool(hr.rand() < 0.15)
        if splits is None and hr.rand() < 0.5:
            spp = int(hr.choice([1, 3, 5, 6, 7, 12]))
    film = scenes.film_desc(xres, yres, fk, crop=crop, **film_kw)
    depth = int(rng.randint(1, 9))
    strategy = [abi.LIGHTS_UNIFORM, abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL][rng.randint(3)]
    rr = float(rng.choice([1.0, 1.0, 0.2, 0.0, 5.0]))
    if integ_kind == "path":
        integ = scenes.integrator_desc(film, spp, max_depth=depth, rr_threshold=rr, light_strategy=strategy, pixel_bounds=pixel_bounds)
    elif integ_kind == "volpath":
        integ = scenes.volpath_desc(film, spp, max_depth=depth, rr_threshold=rr, light_strategy=strategy, pixel_bounds=pixel_bounds)
    else:
        integ = scenes.direct_lighting_desc(film, spp, max_depth=min(depth, 4), strategy=abi.DL_SAMPLE_ALL if integ_kind == "direct-all" else abi.DL_SAMPLE_ONE,
                                            pixel_bounds=pixel_bounds)
    if details is not None:
        details.update(spp=spp, splits=splits, eye=tuple(float(v) for v in eye), look=tuple(float(v) for v in look), up=tuple(float(v) for v in up), fov=fov, xres=int(xres), yres=int(yres),
                       lens=lens, focal=float(cam.focal_distance), filter=str(fk), crop=crop, integ_kind=str(integ_kind), depth=depth, rr=rr, strategy=strategy,
                       environment=environment, ply=ply, sampler=sampler, center=center)
    label = "%s%s%s depth %d rr %g strategy %d filter %s tris %d spheres %d lights %d media %d textures %d%s" % (
        integ_kind, " environment-camera" if environment else "", (" halton%s spp %d" % ("-centre" if center else "", spp)) if sampler == "halton" else "", depth, rr, strategy, fk, sc.indices.shape[0], len(b.spheres), len(b.lights), len(b.media), len(b.textures), (" crop" if crop != (0.0, 1.0, 0.0, 1.0) else "") + (" options %s%s%s" % (sorted(film_kw), pixel_bounds or "", splits or "") if (film_kw or pixel_bounds or splits) else ""))
    return sc, cam, film, integ, label


def channel_floor(ref):
    """The denominator of a relative error per channel: the reference value, at least 1e-3, and at least 1e-6 of the
    pixel's largest channel.  The film holds XYZ (Enigma/Film.cpp:117-120) and the image is its conversion back to RGB:
    under a saturated colour (a pixel of 4e5 in red and blue and nothing in green: seed 1200641, a mix whose weights are 0
    in one channel) the empty channel is what the cancellation of three products of 1e5 leaves, some 1e-7 of them, on
    either side."""
    return np.maximum(np.maximum(ref, 1e-3), 1e-6 * np.abs(ref).max(axis=2, keepdims=True))


def image_verdict(out, a, bimg, base_ok, radius=(4.0, 4.0), rays_apart=0):
    """Fills out["status"] for two images of the same samples.  "ok": the mean relative error is below 1 %.  Otherwise:
    A path through mirror-sharp lobes (alpha 0.001) carries weights of 1e5 and more; a branch flipped by the last bit of a
    sine moves one pixel by orders of magnitude and the mean with it.  Such a firefly is one filter footprint; what a
    defect moves is a material, a light or a shape: many pixels, or the ray counts.  So an image whose mean is off passes
    only as a "firefly": the pixels off by more than 1 % gather into at most four footprints of the film's filter (a
    footprint with negative lobes falls apart into rings and specks: pixels closer than the filter's radius count as one
    patch) -- four, and one more for every ray the two sides' counts differ by, since a path that took another branch
    shows in the counts --, together no more pixels than that many footprints hold, and the mean over all OTHER pixels as
    tight as a clean image's (< 1e-3).  Images
    with values beyond 1e9 (throughputs at the edge of the float range: out["extreme"]) pass as "extreme" on their median
    pixel and ray counts alone.  Both are reported under their names, not as "ok"."""
    from scipy import ndimage
    rel = np.abs(a - bimg).max(axis=2) / np.maximum(np.abs(bimg).max(axis=2), 1e-3)
    off = rel > 0.01
    out["pixels_off"] = float(off.mean())
    ry, rx = int(np.ceil(radius[1])), int(np.ceil(radius[0]))
    joined = ndimage.binary_dilation(off, structure=np.ones((2 * ry + 1, 2 * rx + 1), dtype=bool)) if off.any() else off
    labels, n_patches = ndimage.label(joined, structure=np.ones((3, 3), dtype=int))
    sizes = ndimage.sum(off, labels, index=np.arange(1, n_patches + 1)) if n_patches else np.zeros(0)
    out["patches_off"] = int(n_patches)
    out["largest_patch"] = int(sizes.max()) if n_patches else 0
    footprint = (2 * ry + 2) * (2 * rx + 2)
    rest = ~off
    per_pixel = (np.abs(a - bimg) / channel_floor(bimg)).mean(axis=2)
    out["mre_without_patches"] = float(per_pixel[rest].mean()) if rest.any() else float("inf")
    firefly = n_patches <= 4 + rays_apart and float(off.sum()) <= (4 + rays_apart) * footprint and out["mre_without_patches"] < 1e-3
    if base_ok and out["mre"] < 0.01:
        out["status"] = "ok"
    elif base_ok and out["extreme"]:
        out["status"] = "extreme"
    elif base_ok and firefly:
        out["status"] = "firefly"
    else:
        out["status"] = "mismatch"
    return out


def last_bit_spread(seed, res, spp, base_stats, base_img):
    """The oracle's distance from itself over the six one-unit moves of the camera directions' last place: the largest
    ray-count distance, mean and median relative pixel difference (one process per move: the knob is read once)."""
    import subprocess
    import tempfile
    worst = {"counts_apart": 0.0, "mre": 0.0, "median_rel": 0.0}
    code = ("import sys, numpy as np\nsys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport orc, fuzz_scenes\n"
            "d = {}\nsc, cam, film, integ, label = fuzz_scenes.random_case(%d, %d, %d, details=d)\n"
            "o = orc.Oracle(sc, cam, film, integ, sampler=d['sampler'], sample_at_pixel_center=d['center']); o.render(0, d['spp'], threads=8)\n"
            "np.save(sys.argv[1], o.image()); s = o.stats(); print(s['rays_closest'], s['rays_shadow'], s['rays_mis'])\n") % (
                ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), seed, res, spp)
    with tempfile.TemporaryDirectory() as tmp:
        for knob in range(6):
            path = os.path.join(tmp, "k%d.npy" % knob)
            run = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, ORACLE_CAMERA_LAST_BIT=str(knob)), capture_output=True, text=True, timeout=300)
            if run.returncode != 0:
                return None
            counts = [int(v) for v in run.stdout.split()]
            base = [int(base_stats[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")]
            img = np.load(path)
            rel = np.abs(img - base_img) / channel_floor(base_img)
            worst["counts_apart"] = max(worst["counts_apart"], max(abs(g - c) / max(1000, c) for g, c in zip(counts, base)))
            worst["mre"] = max(worst["mre"], float(rel.mean()))
            worst["median_rel"] = max(worst["median_rel"], float(np.median(rel)))
    return worst


def compare(seed, res=24, spp=4, shared=None):
    """{"status": "ok" | "firefly" | "extreme" | "sensitive" | "both-refuse" | "gpu-refuses" | "oracle-refuses" | "mismatch", ...} for one seed (image_verdict).
    shared: a one-element list holding a Renderer that is kept from seed to seed (scenes, films and integrators of every
    kind configured one after the other on one handle, refusals in between), instead of a fresh handle per seed."""
    import orc
    import raytrace_amd as ra
    import common
    d = {}
    spp_in = spp
    sc, cam, film, integ, label = random_case(seed, res, spp, details=d)
    spp = d["spp"]
    out = {"seed": seed, "label": label}
    r = o = None
    try:
        gerr = oerr = None
        before = None
        try:
            if shared is not None:
                if shared[0] is None:
                    shared[0] = ra.Renderer(0)
                r = shared[0]
                r.configure(sc, cam, film, integ, sampler=d["sampler"], sample_at_pixel_center=d["center"])
                r.clear()
                before = r.stats()
            else:
                r = ra.Renderer(0).configure(sc, cam, film, integ, sampler=d["sampler"], sample_at_pixel_center=d["center"])
            cuts = d["splits"] or [0, spp]
            for a_, b_ in zip(cuts[:-1], cuts[1:]):   # the library in several calls, the oracle in one
                r.render(a_, b_)
        except ra.EnigmaError as e:
            gerr = str(e)
        try:
            o = orc.Oracle(sc, cam, film, integ, sampler=d["sampler"], sample_at_pixel_center=d["center"])
            o.render(0, spp, threads=8)
        except Exception as e:  # the oracle wrapper raises RuntimeError with its message
            oerr = str(e)
        if gerr and not oerr and o.read_past_arrays():
            oerr = "the render read past the integrator's sample arrays (undefined in the reference)"
        if gerr or oerr:
            status = "both-refuse" if (gerr and oerr) else "mismatch"
            # what the library refuses from the scene alone, before a path could run into it (both documented): strategy "all"
            # next to a branching smooth glass (undefined in the reference once a path meets it), an emissive triangle
            # whose vertex normals disagree about its side
            if gerr and not oerr and any(t in gerr for t in DOCUMENTED_REFUSALS):
                status = "gpu-refuses"
            # what the oracle does not restate and refuses before looking at the film, on a film whose tile loop visits no pixel
            # (rows start at the sample bounds' min.x, Integrator.cpp:254: none when that is past max.y), where the library
            # returns the empty render before its own refusal of the same combination: untested here, not a disagreement
            if oerr and not gerr and "are not restated" in oerr:
                status = "oracle-refuses"
            out.update(status=status, gpu_error=gerr, oracle_error=oerr)
            return out
        a, bimg = r.image(), o.image()
        sg, so = r.stats(), o.stats()
        if before is not None:   # the counters of a handle run on across renders
            sg = dict(sg, **{k: sg[k] - before[k] for k in ("rays_closest", "rays_shadow", "rays_mis")})
        counts = {k: (int(sg[k]), int(so[k])) for k in ("rays_closest", "rays_shadow", "rays_mis")}
        out["counts"] = counts
        out["finite"] = bool(np.isfinite(a).all())
        out["mre"] = float(np.mean(np.abs(a - bimg) / channel_floor(bimg)))
        out["median_rel"] = float(np.median(np.abs(a - bimg) / channel_floor(bimg)))
        out["stack_overflows"] = int(sg.get("stack_overflows", 0))
        apart = max(abs(g - c) / max(1000, c) for g, c in counts.values())  # a handful of rays of a few hundred is no percentage
        out["counts_apart"] = apart
        # Shadow rays that carry no weight: where a BSDF value cancels to rounding noise (a conductor with eta = 1, k = 0 has
        # a Fresnel reflectance of 0 in exact arithmetic and of +-1e-9 or exactly 0 in floats, by the last bit of the cosine),
        # "f is black" -- and with it whether a shadow ray is traced at all -- falls either way on either side, while the
        # image cannot tell: the shadow-ray count is then not held to 1 % as long as the images agree to 1e-6 (seed 911751).
        apart_paths = max(abs(g - c) / max(1000, c) for k, (g, c) in counts.items() if k != "rays_shadow")
        out["zero_weight_shadow_rays"] = bool(apart > 0.01 and apart_paths <= 0.01 and out["mre"] < 1e-6)
        if out["zero_weight_shadow_rays"]:
            apart = apart_paths
        # A path whose throughput has run up to 1e7 and beyond (radiance of 1e9 from lights of 1e2) (a mirror-sharp lobe met off its peak, bounce after bounce)
        # sits at the edge of the float range: one more factor overflows or not depending on the last bits of the libm
        # each side uses, and an overflowed sample is scrubbed to zero (Enigma/Integrator.cpp:284-292).  Such images are
        # compared by their ray counts and their median pixel only.
        out["extreme"] = bool(max(float(np.nanmax(np.abs(a))), float(np.nanmax(np.abs(bimg)))) > 1e9)
        image_verdict(out, a, bimg, out["finite"] and out["median_rel"] < 1e-3 and apart <= 0.01 and out["stack_overflows"] == 0,
                      radius=tuple(float(v) for v in film.filter_radius), rays_apart=sum(abs(g - c) for g, c in counts.values()))
        if out["status"] == "mismatch" and out["finite"] and out["stack_overflows"] == 0:
            # Before a difference is called a defect: how far is the oracle from ITSELF when every camera ray's direction is
            # moved by one unit in the last place (tools: ORACLE_CAMERA_LAST_BIT)?  That is the distance two libms put between
            # the two sides wherever a direction goes through sinf / cosf -- the environment camera, a thin lens, a point
            # sampled on a sphere.  In scenes of slivers, mirror-sharp lobes or forward-peaked media a percent or two of the
            # rays, and a third of the pixels, hang on such bits (seeds 1504850, 1609423, 1611671).  A library that is no
            # further from the oracle than twice that is "sensitive", reported under that name.
            spread = last_bit_spread(seed, res, spp_in, so, bimg)
            out["last_bit_spread"] = spread
            if spread and apart <= max(0.01, 2 * spread["counts_apart"]) and out["mre"] <= max(0.01, 2 * spread["mre"]) and \
                    out["median_rel"] <= max(1e-3, 2 * spread["median_rel"]):
                out["status"] = "sensitive"
        return out
    finally:
        if r is not None and shared is None:
            r.close()
        if o is not None:
            o.close()


def adversarial_rays(sc, rng, n=3000):
    """[n, 7] rays (o, d, tMax): uniform ones, and ones built to sit on the decisions of the tests -- aimed exactly at
    vertices, at points on edges and inside triangles, starting on surfaces, along the axes, with zero components, with
    short and enormous tMax."""
    v = sc.vertices.astype(np.float64)
    t = sc.indices
    lo, hi = (v.min(axis=0) - 2, v.max(axis=0) + 2) if len(v) else (np.zeros(3) - 5, np.zeros(3) + 15)
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    tmax = np.full(n, np.inf)
    if len(t):
        k = rng.randint(0, len(t), n)
        tri = v[t[k]]                                    # [n, 3, 3]
        mode = rng.randint(0, 10, n)
        w = rng.dirichlet((1, 1, 1), n)
        inside = (tri * w[:, :, None]).sum(axis=1)
        edge_w = np.stack([rng.rand(n), 1 - rng.rand(n) * 0, np.zeros(n)], axis=1)
        u = rng.rand(n)
        on_edge = tri[:, 0] * u[:, None] + tri[:, 1] * (1 - u)[:, None]
        target = np.where((mode == 0)[:, None], tri[:, 0], np.where((mode == 1)[:, None], on_edge, inside))
        aimed = mode <= 4
        d[aimed] = (target - o)[aimed]
        start_on = mode == 5                              # a ray that starts on a surface
        o[start_on] = inside[start_on]
        ax = mode == 6                                    # axis-aligned, two components exactly zero
        axis = rng.randint(0, 3, n)
        dd = np.zeros((n, 3)); dd[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
        d[ax] = dd[ax]
        zc = mode == 7                                    # one component exactly zero
        d[zc, rng.randint(0, 3)] = 0.0
        short = mode == 8                                 # tMax right at the hit distance
        dist = np.linalg.norm(inside - o, axis=1) / np.maximum(np.linalg.norm(d, axis=1), 1e-30)
        d[short] = (inside - o)[short]
        tmax[short] = rng.choice([1.0, 0.999999, 1.000001, 0.5, 1e-6], int(short.sum()))
        big = mode == 9
        tmax[big] = 3e38
    d[np.abs(d).sum(axis=1) == 0] = (0, 0, 1)
    scale = 10 ** rng.uniform(-3, 3, n)                   # directions are not unit vectors in the reference either
    d = d * scale[:, None]
    tmax = np.where(np.isfinite(tmax) & (tmax < 1e30), tmax / 1.0, tmax)
    return np.concatenate([o, d, tmax[:, None]], axis=1).astype(np.float32)


def compare_rays(seed, n=3000):
    """Closest hits and any-hit decisions of adversarial rays against the seed's scene: primitive, t and barycentrics bit
    for bit (rays the oracle flags as exact ties between two primitives: the same distance)."""
    import orc
    import raytrace_amd as ra
    sc, cam, film, integ, label = random_case(seed)
    out = {"seed": seed, "label": label}
    rng = np.random.RandomState(seed + 77)
    rays = adversarial_rays(sc, rng, n)
    r = o = None
    try:
        try:
            r = ra.Renderer(0)
            r.upload_scene(sc)
        except ra.EnigmaError as e:
            out.update(status="gpu-refuses" if any(t in str(e) for t in DOCUMENTED_REFUSALS) else "mismatch", gpu_error=str(e))
            return out
        o = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, 4))
        ho, hg = o.intersect(rays), r.intersect(rays)
        ok = ho["ambiguous"] == 0
        u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
        on_triangle = ho["tri"] < sc.indices.shape[0]    # for a sphere the taps carry different things in place of barycentrics
        bad = (ho["prim_id"] != hg["prim_id"]) | (ho["tri"] != hg["tri"]) | (u32(ho["t"]) != u32(hg["t"])) | ((u32(ho["bary"]) != u32(hg["bary"])).any(axis=1) & on_triangle)
        bad &= ok
        amb_bad = (~ok) & ~np.isclose(ho["t"], hg["t"], rtol=2e-6)
        any_o, any_g = o.intersect_any(rays), r.intersect_any(rays)
        any_bad = (any_o != any_g) & ok
        out.update(rays=int(n), hits=int((ho["tri"] >= 0).sum()), ambiguous=int((~ok).sum()), closest_bad=int(bad.sum()), tie_bad=int(amb_bad.sum()), any_bad=int(any_bad.sum()))
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            out["first_bad"] = {"ray": rays[i].tolist(), "oracle": [int(ho["tri"][i]), float(ho["t"][i])] + ho["bary"][i].tolist(), "gpu": [int(hg["tri"][i]), float(hg["t"][i])] + hg["bary"][i].tolist()}
        if any_bad.any():
            i = int(np.nonzero(any_bad)[0][0])
            out["first_any_bad"] = {"ray": rays[i].tolist(), "oracle": int(any_o[i]), "gpu": int(any_g[i])}
        out["status"] = "ok" if not (bad.any() or amb_bad.any() or any_bad.any()) else "mismatch"
        return out
    finally:
        if r is not None:
            r.close()
        if o is not None:
            o.close()


def adversarial_directions(rng, n):
    """(wo, wi, u): unit and non-unit directions all over the sphere, many of them on the decisions of the BSDF code
    (grazing, exactly in the surface, along the normal, mirror pairs, opposite hemispheres), sample values at the ends
    of their range."""
    def dirs():
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1)[:, None]
        k = rng.randint(0, 8, n)
        d[k == 0, 2] = 0.0                                           # exactly in the surface
        d[k == 1, 2] = rng.choice([1e-4, -1e-4, 1e-7, -1e-7], int((k == 1).sum()))   # grazing
        d[k == 2] = (0.0, 0.0, 1.0)
        d[k == 3] = (0.0, 0.0, -1.0)
        nz = np.linalg.norm(d, axis=1)
        nz[nz == 0] = 1
        return d / nz[:, None], k
    wo, _ = dirs()
    wi, k = dirs()
    mirror = rng.rand(n) < 0.15
    wi[mirror] = wo[mirror] * (-1, -1, 1)                            # the mirror direction: half vector along the normal
    through = rng.rand(n) < 0.1
    wi[through] = -wo[through]                                       # straight through
    u = rng.rand(n, 2)
    e = rng.rand(n) < 0.2
    u[e] = rng.choice([0.0, 0.99999994, 0.5, 1e-7], (int(e.sum()), 2))
    return wo.astype(np.float32), wi.astype(np.float32), u.astype(np.float32)


def compare_bsdf(seed, n=4000):
    """BSDF::f, Pdf and sample_f of a random material at adversarial (wo, wi, u): the device's evaluation tap against the
    oracle's.  Sampled lobe types must be equal; values within 2e-4 relative (libm), with an absolute floor of 1e-5 of
    the largest value of the column (cancellation near zeros of the Fresnel and masking terms); non-finite values and
    exact zeros must coincide."""
    import orc
    import raytrace_amd as ra
    rng = np.random.RandomState(seed)
    m = random_material(rng)
    if rng.rand() < 0.3:   # the edges of the parameter ranges
        k = rng.randint(5)
        if k == 0: m.uroughness = m.vroughness = 0.0
        elif k == 1: m.index = float(rng.choice([1.0, 1.0001, 3.0, 0.9]))
        elif k == 2: m.kd[:] = [0, 0, 0]; m.ks[:] = [0, 0, 0]; m.kr[:] = [0, 0, 0]
        elif k == 3: m.kt[:] = [0, 0, 0]
        else: m.uroughness, m.vroughness = float(rng.uniform(1, 3)), float(rng.uniform(0.001, 3))
    wo, wi, u = adversarial_directions(rng, n)
    r = ra.Renderer(0)
    try:
        got, want = r.bsdf_eval(m, wo, wi, u), orc.bsdf_eval(m, wo, wi, u)
    finally:
        r.close()
    out = {"seed": seed, "material": [int(m.type), float(m.uroughness), float(m.vroughness), float(m.index), float(m.sigma), int(m.remap_roughness)]}
    np.seterr(all="ignore")   # the differences of values that are not finite on either side are masked below, not computed around
    fin_g, fin_w = np.isfinite(got[:, :11]), np.isfinite(want[:, :11])
    lobes = got[:, 11] != want[:, 11]
    both = fin_g & fin_w
    col_scale = np.maximum(np.nanmax(np.where(both, np.abs(want[:, :11]), 0), axis=0), 1e-30)
    err = np.where(both, np.abs(got[:, :11] - want[:, :11]) / np.maximum(np.abs(want[:, :11]), 1e-5 * col_scale[None, :]), 0)
    zero = ((got[:, :11] == 0) != (want[:, :11] == 0)) & both & (np.maximum(np.abs(got[:, :11]), np.abs(want[:, :11])) > 1e-5 * col_scale[None, :])
    # a direction sampled at the rim of the hemisphere: z = sqrt(1 - x^2 - y^2) cancels to the last bits of the cosine and
    # sine of the two libms (0.00065 against 0.00055 at u = 1e-7); the sampled columns of such rows are not compared
    rim = np.abs(want[:, 10]) < 3e-3
    err[rim, 4:11] = 0
    zero[rim, 4:11] = False
    # away from the rim the same cancellation costs 2.4e-7 / z^2 of relative accuracy: the sampled columns get that much more room
    room = 2.4e-7 / np.maximum(want[:, 10] ** 2, 1e-12)
    # a sharp microfacet lobe magnifies the last bits of the sampled direction (the two libms' sine and cosine: some 1e-6
    # rad) in its value and density: D falls off over an angle alpha, so a step d(theta) changes it by about
    # 4 tan(theta) d(theta) / alpha^2 of itself -- 7e-4 at alpha = 0.01, seen at seed 900080, identical in -O1 and -O3 builds
    alpha = min(float(m.uroughness), float(m.vroughness))
    if alpha > 0:
        room = room + 8e-8 / alpha ** 2
    err[:, 4:11] = np.maximum(err[:, 4:11] - room[:, None], 0)
    if float(m.index) == 1.0:
        # a dielectric with the index of its surroundings: the refracted direction is -wo up to rounding, and the
        # transmission lobe divides by (wo.wh + eta wi.wh)^2, which is 0 in exact arithmetic -- 0, 1e-16 or NaN in floats, by
        # the last bit of a square root (the reference's own arithmetic, Enigma/BxDF.cpp; seed 910233).  Rows whose sampled
        # lobe is a transmission on either side are not compared.
        through = ((got[:, 11].astype(np.int64) | want[:, 11].astype(np.int64)) & 2) != 0
        lobes = lobes & ~through
        fin_g = np.where(through[:, None], True, fin_g); fin_w = np.where(through[:, None], True, fin_w)
        err[through, 4:11] = 0
        zero[through, 4:11] = False
    rows = lobes | (fin_g != fin_w).any(axis=1) | (err > 2e-4).any(axis=1) | zero.any(axis=1)
    out.update(points=int(n), lobe_diff=int(lobes.sum()), finite_diff=int((fin_g != fin_w).any(axis=1).sum()), value_diff=int((err > 2e-4).any(axis=1).sum()),
               zero_diff=int(zero.any(axis=1).sum()), max_err=float(err.max()))
    if rows.any():
        i = int(np.nonzero(rows)[0][0])
        out["first"] = {"wo": wo[i].tolist(), "wi": wi[i].tolist(), "u": u[i].tolist(), "gpu": got[i].tolist(), "oracle": want[i].tolist()}
    out["status"] = "ok" if not rows.any() else "mismatch"
    return out


def compare_taps(seed):
    """The bit-exact tiers at a random film: Sobol' sample values and camera rays (pinhole) of random pixels and sample
    numbers, film footprints (pixel indices and filter weights) of random and lattice-aligned film positions -- for a
    random resolution up to 4096, crop window, pixel filter and radii, samples per pixel up to 2^16."""
    import orc
    import raytrace_amd as ra
    rng = np.random.RandomState(seed)
    xres, yres = int(rng.choice([1, 2, 3, 17, 64, 100, 257, 640, 1000, 1920, 4096])), int(rng.choice([1, 2, 5, 31, 64, 99, 256, 480, 1080, 2160, 4096]))
    fk = str(rng.choice(["box", "triangle", "gaussian", "mitchell", "sinc"]))
    kw = {}
    if rng.rand() < 0.5:
        kw["radius"] = (float(rng.uniform(0.3, 4.5)), float(rng.uniform(0.3, 4.5)))
    crop = (0.0, 1.0, 0.0, 1.0)
    if rng.rand() < 0.4:
        x0, y0 = rng.uniform(0, 0.6, 2)
        crop = (float(x0), float(min(1.0, x0 + rng.uniform(0.1, 0.6))), float(y0), float(min(1.0, y0 + rng.uniform(0.1, 0.6))))
    film = scenes.film_desc(xres, yres, fk, crop=crop, **kw)
    spp = int(2 ** rng.randint(0, 17))
    integ = scenes.integrator_desc(film, spp, max_depth=int(rng.randint(1, 9)))
    eye, look = rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3)
    if np.linalg.norm(look - eye) < 1e-2:
        look = eye + (0, 0, 1)
    cam = scenes.camera_desc(tuple(eye), tuple(look), (0.1, 1.0, 0.05), float(rng.uniform(10, 140)), xres, yres)
    sc = scenes.cornell_box("diffuse")
    # seeds from 1,700,000 on: the HaltonSampler for half of the films, with any sample count up to 2^16 (drawn last: the films of the seeds before stay)
    sampler, center = "sobol", False
    if seed >= 1700000 and rng.rand() < 0.5:
        sampler, center = "halton", bool(rng.rand() < 0.1)
        spp = int(rng.randint(1, 65537))
        integ = scenes.integrator_desc(film, spp, max_depth=int(integ.max_depth))
    out = {"seed": seed, "label": "%dx%d crop %s filter %s %s spp %d%s" % (xres, yres, crop, fk, kw.get("radius"), spp, " halton" + ("-centre" if center else "") if sampler == "halton" else "")}
    r = o = None
    try:
        gerr = oerr = None
        try:
            r = ra.Renderer(0).configure(sc, cam, film, integ, sampler=sampler, sample_at_pixel_center=center)
        except ra.EnigmaError as e:
            gerr = str(e)
        try:
            o = orc.Oracle(sc, cam, film, integ, sampler=sampler, sample_at_pixel_center=center)
            sb = scenes.sample_bounds(film)
            if sb[0] < sb[1]:
                oerr = "sample bounds with min.x < min.y: reference result undefined"
        except Exception as e:
            oerr = str(e)
        if gerr or oerr:
            out.update(status="both-refuse" if (gerr and oerr) else ("gpu-refuses" if gerr and ("min.x < min.y" in gerr or "Log2Int" in gerr or "empty sample bounds" in gerr) else "mismatch"), gpu_error=gerr, oracle_error=oerr)
            # (the library looks at the sample bounds when the sampler is configured, the oracle when it renders)
            if not gerr and oerr and "min.x < min.y" in oerr:
                out["status"] = "both-refuse"
            return out
        sb = scenes.sample_bounds(film)
        n = 2000
        P = np.stack([rng.randint(sb[0], max(sb[0] + 1, sb[2]), n), rng.randint(sb[1], max(sb[1] + 1, sb[3]), n)], axis=1).astype(np.int32)
        P[:4] = [(sb[0], sb[1]), (sb[2] - 1, sb[3] - 1), (sb[2] - 1, sb[1]), (sb[0], sb[3] - 1)]
        S = rng.randint(0, spp, n).astype(np.int64)
        S[:3] = [0, spp - 1, spp // 2]
        u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
        nd = 5 + 8 * int(integ.max_depth)                            # the dimensions a path of this depth can consume are tabulated
        io, do = o.sobol(P, S, nd)
        ig, dg = r.sobol(P, S, nd)
        ro, po = o.camera_rays(P, S)
        rg, pg = r.camera_rays(P, S)
        b = film.cropped_pixel_bounds
        pf = np.stack([rng.uniform(b[0] - 5, b[2] + 5, n), rng.uniform(b[1] - 5, b[3] + 5, n)], axis=1).astype(np.float32)
        lat = rng.rand(n) < 0.3                                      # positions on the half-pixel lattice, where ceil / floor decide
        pf[lat] = np.round(pf[lat] * 2) / 2
        fo, fg = o.film_footprint(pf, 100), r.film_footprint(pf, 100)
        used = np.arange(100)[None, :] < np.minimum(fo[2], 100)[:, None]      # entries past a position's count are not written
        fo = (np.where(used[:, :, None], fo[0], 0), np.where(used, fo[1], 0), fo[2])
        fg = (np.where(used[:, :, None], fg[0], 0), np.where(used, fg[1], 0), fg[2])
        out.update(sobol_index=int((io != ig).sum()), sobol_bits=int((u32(do) != u32(dg)).sum()), ray_bits=int((u32(ro) != u32(rg)).sum()), pfilm_bits=int((u32(po) != u32(pg)).sum()),
                   footprint_counts=int((fo[2] != fg[2]).sum()), footprint_pixels=int((fo[0] != fg[0]).sum()), footprint_weights=int((u32(fo[1]) != u32(fg[1])).sum()))
        bad = sum(out[k] for k in ("sobol_index", "sobol_bits", "ray_bits", "pfilm_bits", "footprint_counts", "footprint_pixels", "footprint_weights"))
        out["status"] = "ok" if bad == 0 else "mismatch"
        return out
    finally:
        if r is not None:
            r.close()
        if o is not None:
            o.close()


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return img[::-1]


def compare_file(seed, directory, res=24, spp=4):
    """The same comparison through the scene-description front end: the seed's scene written as pbrt-format text
    (tests/pbrt_writer.py), rendered by raytrace_amd/enigma_render (front end, host mirror classes, C ABI, GPU), against
    the oracle rendering the scene built in Python."""
    import subprocess
    import orc
    import common
    import pbrt_writer
    d = {}
    sc, cam, film, integ, label = random_case(seed, res, spp, for_file=True, details=d)
    spp = d["spp"]   # the sampler stream may have drawn another count
    out = {"seed": seed, "label": label}
    strategy = {abi.LIGHTS_UNIFORM: "uniform", abi.LIGHTS_POWER: "power", abi.LIGHTS_SPATIAL: "spatial"}[d["strategy"]]
    if d["integ_kind"] == "path":
        itext = 'Integrator "path" "integer maxdepth" %d "float rrthreshold" %.9g "string lightsamplestrategy" "%s"' % (d["depth"], d["rr"], strategy)
    elif d["integ_kind"] == "volpath":
        itext = 'Integrator "volpath" "integer maxdepth" %d "float rrthreshold" %.9g "string lightsamplestrategy" "%s"' % (d["depth"], d["rr"], strategy)
    else:
        itext = 'Integrator "directlighting" "integer maxdepth" %d "string strategy" "%s"' % (min(d["depth"], 4), "all" if d["integ_kind"] == "direct-all" else "one")
    crop = d["crop"]
    film_extra = "" if crop == (0.0, 1.0, 0.0, 1.0) else '"float cropwindow" [%.9g %.9g %.9g %.9g]' % crop
    lens = '"float lensradius" %.9g "float focaldistance" %.9g' % (d["lens"], d["focal"]) if d["lens"] else ""
    text = pbrt_writer.scene_text(sc, d["eye"], d["look"], d["up"], d["fov"], d["xres"], d["yres"], spp, integrator=itext,
                                  pixel_filter='PixelFilter "%s"' % d["filter"], film_extra=film_extra, lens=lens, texture_dir=directory,
                                  ply_dir=directory if d["ply"] else None, camera="environment" if d["environment"] else "perspective",
                                  sampler='Sampler "sobol"' if d["sampler"] == "sobol" else 'Sampler "halton"' + (' "bool samplepixelcenter" "true"' if d["center"] else ""))
    scene_file, out_file = os.path.join(directory, "s%d.pbrt" % seed), os.path.join(directory, "s%d.pfm" % seed)
    with open(scene_file, "w") as f:
        f.write(text)
    p = subprocess.run([os.path.join(ROOT, "raytrace_amd", "enigma_render"), scene_file, out_file], capture_output=True, text=True, timeout=300)
    gerr = None if p.returncode == 0 else (p.stderr.strip().splitlines() or ["exit status %d" % p.returncode])[-1]
    if p.returncode < 0 or p.returncode > 2:
        out.update(status="mismatch", gpu_error="enigma_render died: %s" % gerr)
        return out
    o = oerr = None
    try:
        o = orc.Oracle(sc, cam, film, integ, sampler=d["sampler"], sample_at_pixel_center=d["center"])
        o.render(0, spp, threads=8)
    except Exception as e:
        oerr = str(e)
    try:
        if gerr and not oerr and o.read_past_arrays():
            oerr = "the render read past the integrator's sample arrays (undefined in the reference)"
        if gerr or oerr:
            status = "both-refuse" if (gerr and oerr) else "mismatch"
            if gerr and not oerr and any(t in gerr for t in DOCUMENTED_REFUSALS):
                status = "gpu-refuses"
            if oerr and not gerr and "are not restated" in oerr:
                status = "oracle-refuses"
            out.update(status=status, gpu_error=gerr, oracle_error=oerr)
            return out
        a, bimg = read_pfm(out_file), o.image()
        if a.shape != bimg.shape:
            out.update(status="mismatch", gpu_error="image %s against %s" % (a.shape, bimg.shape))
            return out
        out["finite"] = bool(np.isfinite(a).all())
        out["mre"] = float(np.mean(np.abs(a - bimg) / channel_floor(bimg)))
        out["median_rel"] = float(np.median(np.abs(a - bimg) / channel_floor(bimg)))
        out["extreme"] = bool(max(float(np.nanmax(np.abs(a))), float(np.nanmax(np.abs(bimg)))) > 1e9)
        return image_verdict(out, a, bimg, out["finite"] and out["median_rel"] < 1e-3)
    finally:
        if o is not None:
            o.close()
        for f in (scene_file, out_file):
            if os.path.exists(f) and out.get("status") != "mismatch":
                os.remove(f)


def main():
    if sys.argv[1] == "--taps":    # python tools/fuzz_scenes.py --taps <first seed> <count>
        first, count = int(sys.argv[2]), int(sys.argv[3])
        results = []
        for seed in range(first, first + count):
            results.append(compare_taps(seed))
            if results[-1]["status"] == "mismatch":
                print(json.dumps(results[-1]), flush=True)
            elif seed % 20 == 0:
                print("seed %d %s" % (seed, results[-1]["status"]), flush=True)
        import collections
        print("taps, seeds %d..%d: %s" % (first, first + count - 1, dict(collections.Counter(r["status"] for r in results))))
        sys.exit(1 if any(r["status"] == "mismatch" for r in results) else 0)
    if sys.argv[1] == "--bsdf":    # python tools/fuzz_scenes.py --bsdf <first seed> <count>
        first, count = int(sys.argv[2]), int(sys.argv[3])
        results = []
        for seed in range(first, first + count):
            results.append(compare_bsdf(seed))
            if results[-1]["status"] == "mismatch":
                print(json.dumps(results[-1]), flush=True)
            elif seed % 50 == 0:
                print("seed %d %s" % (seed, results[-1]["status"]), flush=True)
        print("bsdf, seeds %d..%d: %d ok, %d mismatch; largest relative difference among the ok ones %.3g" % (
            first, first + count - 1, sum(r["status"] == "ok" for r in results), sum(r["status"] == "mismatch" for r in results),
            max([r["max_err"] for r in results if r["status"] == "ok"] or [0])))
        sys.exit(1 if any(r["status"] == "mismatch" for r in results) else 0)
    if sys.argv[1] == "--rays":    # python tools/fuzz_scenes.py --rays <first seed> <count>
        first, count = int(sys.argv[2]), int(sys.argv[3])
        results = []
        for seed in range(first, first + count):
            results.append(compare_rays(seed))
            if results[-1]["status"] == "mismatch":
                print(json.dumps(results[-1]), flush=True)
            elif seed % 50 == 0:
                print("seed %d %s" % (seed, results[-1]["status"]), flush=True)
        import collections
        print("rays, seeds %d..%d: %s; %d rays, %d hits, %d exact ties" % (first, first + count - 1, dict(collections.Counter(r["status"] for r in results)),
              sum(r.get("rays", 0) for r in results), sum(r.get("hits", 0) for r in results), sum(r.get("ambiguous", 0) for r in results)))
        sys.exit(1 if any(r["status"] == "mismatch" for r in results) else 0)
    if sys.argv[1] == "--files":   # python tools/fuzz_scenes.py --files <first seed> <count> <directory>
        first, count, directory = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        os.makedirs(directory, exist_ok=True)
        results = [compare_file(seed, directory) for seed in range(first, first + count)]
        for r in results:
            if r["status"] != "ok":
                print(json.dumps(r), flush=True)
        import collections
        print("scene files, seeds %d..%d: %s" % (first, first + count - 1, dict(collections.Counter(r["status"] for r in results))))
        sys.exit(1 if any(r["status"] == "mismatch" for r in results) else 0)
    first, count = int(sys.argv[1]), int(sys.argv[2])
    shared = [None] if os.environ.get("FUZZ_REUSE") else None   # FUZZ_REUSE=1: one handle for all the seeds
    res_px = int(os.environ.get("FUZZ_RES", "24"))   # FUZZ_RES=256: films of 256 to 264 pixels a side (many chunks and blocks per launch)
    results, bad = [], 0
    for seed in range(first, first + count):
        res = compare(seed, res=res_px, shared=shared)
        results.append(res)
        if res["status"] == "mismatch":
            bad += 1
        print(json.dumps(res), flush=True)
    if len(sys.argv) > 3:
        json.dump(results, open(sys.argv[3], "w"), indent=1)
    print("seeds %d..%d: %d ok, %d firefly, %d extreme, %d sensitive, %d both refuse, %d refused by the library alone (documented), %d not restated by the oracle, %d mismatch" % (
        first, first + count - 1, sum(r["status"] == "ok" for r in results), sum(r["status"] == "firefly" for r in results), sum(r["status"] == "extreme" for r in results),
        sum(r["status"] == "sensitive" for r in results), sum(r["status"] == "both-refuse" for r in results),
        sum(r["status"] == "gpu-refuses" for r in results), sum(r["status"] == "oracle-refuses" for r in results), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
