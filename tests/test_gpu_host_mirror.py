"""GPU test of the C++ host mirror (raytrace_amd/csrc/host/enigma_host.h): a program written against
the reference's class interface (Film, PerspectiveCamera, SobolSampler, Scene, Integrator::Render)
renders through the C ABI and its Film::WriteImage output matches the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

import common
import orc
from raytrace_amd import abi, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytrace_amd", "enigma_render")


def write_scene(path, sc, cam, xres, yres, spp, max_depth, strategy, rr=1.0):
    nT, nV = sc.indices.shape[0], sc.vertices.shape[0]
    emis = (sc.tri_light >= 0).astype(np.int32)
    Le = np.zeros((nT, 3), dtype=np.float32)
    for t in range(nT):
        if sc.tri_light[t] >= 0:
            Le[t] = sc.lights[sc.tri_light[t]].L[:]
    with open(path, "wb") as f:
        f.write(b"ESC1")
        f.write(struct.pack("<8i", xres, yres, spp, max_depth, strategy, nV, nT, sc.n_materials))
        f.write(struct.pack("<3f", rr, cam.lens_radius, cam.focal_distance))
        f.write(np.array(cam.raster_to_camera[:], dtype=np.float32).tobytes())
        f.write(np.array(cam.camera_to_world[:], dtype=np.float32).tobytes())
        f.write(sc.vertices.astype(np.float32).tobytes())
        f.write(sc.indices.astype(np.int32).tobytes())
        f.write(sc.tri_material.astype(np.int32).tobytes())
        f.write(emis.tobytes())
        f.write(Le.tobytes())
        f.write(bytes(sc.materials))


@pytest.mark.parametrize("variant", ["diffuse", "microfacet"])
def test_host_mirror_renders_like_the_reference_path(tmp_path, variant):
    assert os.path.exists(EXE), "raytrace_amd/enigma_render is not built (make -C raytrace_amd/csrc)"
    res, spp = 64, 16
    sc, cam, film, integ = common.cornell_setup(res=res, spp=spp, variant=variant)
    scene_file, out_file = str(tmp_path / "scene.bin"), str(tmp_path / "out.f32")
    write_scene(scene_file, sc, cam, res, res, spp, 5, 1)
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    img = np.fromfile(out_file, dtype=np.float32).reshape(res, res, 3)
    o = orc.Oracle(sc, cam, film, integ)
    o.render(0, spp, threads=4)
    assert common.mean_relative_error(img, o.image()) < 1e-4
    o.close()


@pytest.mark.parametrize("strategy,code", [(abi.DL_SAMPLE_ALL, 2), (abi.DL_SAMPLE_ONE, 3)])
def test_host_mirror_direct_lighting(tmp_path, strategy, code):
    """GpuDirectLightingIntegrator behind the mirrored SamplerIntegrator::Render."""
    assert os.path.exists(EXE), "raytrace_amd/enigma_render is not built (make -C raytrace_amd/csrc)"
    res, spp = 48, 8
    sc, cam, film, _ = common.cornell_setup(res=res, spp=spp, variant="microfacet")
    scene_file, out_file = str(tmp_path / "scene.bin"), str(tmp_path / "out.f32")
    write_scene(scene_file, sc, cam, res, res, spp, 3, code)
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    img = np.fromfile(out_file, dtype=np.float32).reshape(res, res, 3)
    o = orc.Oracle(sc, cam, film, scenes.direct_lighting_desc(film, spp, max_depth=3, strategy=strategy))
    o.render(0, spp, threads=4)
    assert common.mean_relative_error(img, o.image()) < 1e-4
    o.close()


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0                      # little endian
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return img[::-1]                                        # PFM stores the bottom row first


@pytest.mark.parametrize("variant,integrator", [("microfacet", "path"), ("specular", "path"), ("diffuse", "directlighting"), ("diffuse+delta", "path"), ("fog:microfacet", "volpath"), ("sky:specular", "path"), ("skymap:microfacet", "path"), ("smooth", "path"), ("textured", "path"), ("spheres", "path"), ("spheres+fog", "volpath"), ("uber", "path"), ("mix", "path"), ("fog:mix", "volpath"), ("ply:smooth", "path"), ("ply:textured", "path"), ("textured_lobes", "path"), ("diffuse+projection", "path"), ("diffuse+goniometric", "path")])
def test_scene_description_file_renders_like_the_oracle(tmp_path, variant, integrator):
    """pbrt-format text -> front end -> GpuPathIntegrator / GpuDirectLightingIntegrator -> PFM, against the oracle
    rendering the same scene built in Python."""
    import math
    import pbrt_writer
    assert os.path.exists(EXE), "raytrace_amd/enigma_render is not built (make -C raytrace_amd/csrc)"
    xres, yres, spp = 72, 48, 8
    ply_dir = None
    if variant.startswith("ply:"):       # the meshes in PLY files (ascii and both binary byte orders in turn) under Shape "plymesh"
        variant, ply_dir = variant.split(":")[1], str(tmp_path)
    if variant.startswith("fog:"):
        sc = scenes.cornell_fog(variant.split(":")[1])
    elif variant.startswith("skymap:"):
        sc = scenes.cornell_sky(variant.split(":")[1], radiance_map=True)       # LightSource "infinite" "string mapname": the radiance map in a PFM file
    elif variant.startswith("sky:"):
        sc = scenes.cornell_sky(variant.split(":")[1], radiance_map=False)      # a constant environment
    elif variant == "textured_lobes":
        sc = scenes.cornell_textured_lobes(True)   # "texture Kd" / "texture Ks" / "texture Kr" on uber, substrate, mirror, translucent, a matte inside a mix
    elif variant == "textured":
        sc = scenes.cornell_textured(True)  # Texture "imagemap" directives over PFM files, "texture Kd" bindings, uvs
    elif variant == "spheres+fog":
        sc = scenes.cornell_spheres(fog=True, lamp=True)   # a medium inside the glass sphere: MediumInterface before Shape "sphere"; AreaLightSource on a sphere
    elif variant == "spheres":
        sc = scenes.cornell_spheres()       # Shape "sphere" under ConcatTransform, ReverseOrientation, zmin / zmax / phimax
    elif variant == "smooth":
        sc = scenes.cornell_smooth()       # per-vertex normals, uvs and ReverseOrientation through "normal N" / "float uv"
    else:
        extra = [abi.projection_light((150.0, 400.0, 150.0), (6.0e5, 4.0e5, 3.0e5), fov=60.0)] if variant.endswith("+projection") else []   # LightSource "projection" under Translate
        if variant.endswith("+goniometric"):     # LightSource "goniometric" with its map in a PFM file
            extra = [abi.goniometric_light((150.0, 400.0, 150.0), (4.0e5, 4.0e5, 4.0e5), scenes.sky_map(12, 20))]
        sc = scenes.cornell_box(variant.split("+")[0], delta_lights=variant.endswith("+delta"), extra_lights=extra)
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))
    itext = {"path": 'Integrator "path" "integer maxdepth" 5 "string lightsamplestrategy" "power"',
             "volpath": 'Integrator "volpath" "integer maxdepth" 5 "string lightsamplestrategy" "power"',
             "directlighting": 'Integrator "directlighting" "integer maxdepth" 5 "string strategy" "all"'}[integrator]
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), fov, xres, yres, spp, integrator=itext, texture_dir=str(tmp_path), ply_dir=ply_dir)
    assert ply_dir is None or 'Shape "plymesh"' in text
    scene_file, out_file = str(tmp_path / "cornell.pbrt"), str(tmp_path / "out.pfm")
    with open(scene_file, "w") as f:
        f.write(text)
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    img = _read_pfm(out_file)
    cam, film = scenes.cornell_camera(xres, yres), scenes.film_desc(xres, yres)
    integ = {"path": scenes.integrator_desc, "volpath": scenes.volpath_desc, "directlighting": scenes.direct_lighting_desc}[integrator](film, spp)
    o = orc.Oracle(sc, cam, film, integ)
    o.render(0, spp, threads=4)
    # the front end evaluates the camera matrices in double and rounds once; the Python builder does the same
    assert common.mean_relative_error(img, o.image()) < (5e-3 if integrator == "volpath" else 1e-3)
    o.close()


def test_scene_description_default_light_strategy_is_spatial(tmp_path):
    """lightsamplestrategy defaults to "spatial" (Enigma/PathIntegrator.cpp:199): a scene file that does not name a strategy
    renders like the oracle under ENIGMA_LIGHTS_SPATIAL."""
    import math
    import pbrt_writer
    sc = scenes.cornell_box("diffuse")
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), fov, 48, 48, 8, integrator='Integrator "path"')
    scene_file, out_file = str(tmp_path / "c.pbrt"), str(tmp_path / "o.pfm")
    with open(scene_file, "w") as f:
        f.write(text)
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    cam, film = scenes.cornell_camera(48, 48), scenes.film_desc(48, 48)
    o = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, 8, light_strategy=abi.LIGHTS_SPATIAL))
    o.render(0, 8, threads=4)
    assert common.mean_relative_error(_read_pfm(out_file), o.image()) < 1e-3
    o.close()


def test_scene_file_with_the_default_metal_renders_like_the_oracle_with_copper(tmp_path):
    """Material "metal" that names neither eta nor k (CreateMetalMaterial's copper, roughness 0.5, remapped,
    Enigma/Material.cpp:604-613): the scene file through enigma_render against the oracle given the oracle's own
    RGBSpectrum::FromSampled of the copper tables."""
    import math
    import pbrt_writer
    xres, yres, spp = 72, 48, 8
    sc = scenes.cornell_box("microfacet")
    eta, k = orc.default_metal()
    idx = [i for i in range(sc.n_materials) if sc.materials[i].type == abi.MAT_METAL]
    assert idx
    for i in idx:
        sc.materials[i] = abi.metal([float(v) for v in eta], [float(v) for v in k], roughness=0.5, remap=True)
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), fov, xres, yres, spp)
    lines = text.split("\n")
    n_default = 0
    for j, ln in enumerate(lines):
        if '"metal"' in ln:    # drop every parameter: the front end must come up with the same material by itself
            head = ln[:ln.index('"metal"') + len('"metal"')]
            lines[j] = head
            n_default += 1
    assert n_default == len(idx)
    scene_file, out_file = str(tmp_path / "copper.pbrt"), str(tmp_path / "out.pfm")
    with open(scene_file, "w") as f:
        f.write("\n".join(lines))
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    img = _read_pfm(out_file)
    cam, film = scenes.cornell_camera(xres, yres), scenes.film_desc(xres, yres)
    o = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, spp))
    o.render(0, spp, threads=4)
    assert common.mean_relative_error(img, o.image()) < 1e-3
    o.close()


def test_scene_description_with_the_environment_camera(tmp_path):
    """Camera "environment" in a scene file (CreateEnvironmentCamera, Enigma/Camera.cpp:476-510) against the oracle."""
    import pbrt_writer
    sc = scenes.cornell_box("microfacet")
    eye, look, up = (278.0, 273.0, 279.6), (278.0, 273.0, 600.0), (0.0, 1.0, 0.0)
    text = pbrt_writer.scene_text(sc, eye, look, up, 40.0, 64, 32, 8)
    lines = text.split("\n")
    k = [i for i, l in enumerate(lines) if l.startswith('Camera "perspective"')]
    assert len(k) == 1
    lines[k[0]] = 'Camera "environment"'
    scene_file, out_file = str(tmp_path / "env.pbrt"), str(tmp_path / "env.pfm")
    with open(scene_file, "w") as f:
        f.write("\n".join(lines))
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    film = scenes.film_desc(64, 32)
    o = orc.Oracle(sc, scenes.environment_camera_desc(eye, look, up), film, scenes.integrator_desc(film, 8))
    o.render(0, 8, threads=4)
    assert common.mean_relative_error(_read_pfm(out_file), o.image()) < 1e-3
    o.close()


@pytest.mark.parametrize("line,spp,center", [('Sampler "halton"', 6, False), ('Sampler "halton" "bool samplepixelcenter" "true"', 5, True), (None, 16, False)])
def test_scene_description_with_the_halton_sampler(tmp_path, line, spp, center):
    """Sampler "halton" (CreateHaltonSampler, Enigma/Sampler.cpp:521-528), and no Sampler directive at all, which is the same
    sampler with 16 samples (Enigma/Api.h:82): the scene file through enigma_render against the oracle's HaltonSampler."""
    import math
    import pbrt_writer
    sc = scenes.cornell_box("microfacet")
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), fov, 60, 44, spp, sampler=line)
    assert ('\nSampler "' in text) == (line is not None)
    scene_file, out_file = str(tmp_path / "h.pbrt"), str(tmp_path / "h.pfm")
    with open(scene_file, "w") as f:
        f.write(text)
    p = subprocess.run([EXE, scene_file, out_file], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    cam, film = scenes.cornell_camera(60, 44), scenes.film_desc(60, 44)
    o = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, spp), sampler="halton", sample_at_pixel_center=center)
    o.render(0, spp, threads=4)
    assert common.mean_relative_error(_read_pfm(out_file), o.image()) < 1e-3
    o2 = orc.Oracle(sc, cam, film, scenes.integrator_desc(film, spp))            # not what the Sobol' sampler renders
    o2.render(0, spp, threads=4)
    assert common.mean_relative_error(_read_pfm(out_file), o2.image()) > 0.02
    o.close(); o2.close()
