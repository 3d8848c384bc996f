"""Scene-description front end (raytrace_amd/csrc/host/pbrt_frontend.h), CPU part: the driver's --dump mode parses
a pbrt-format file and writes what it would hand to the GPU integrator; no GPU is touched.  The expected values
come from the Python scene builders (the same arrays the parity tests render) and from first principles."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import common
import pbrt_writer
from raytrace_amd import abi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytrace_amd", "enigma_render")
FOV = 2 * math.degrees(math.atan(0.0125 / 0.035))


def dump(tmp_path, text, expect_fail=False):
    assert os.path.exists(EXE), "raytrace_amd/enigma_render is not built (make -C raytrace_amd/csrc)"
    src, out = str(tmp_path / "scene.pbrt"), str(tmp_path / "scene.json")
    with open(src, "w") as f:
        f.write(text)
    p = subprocess.run([EXE, "--dump", src, out], capture_output=True, text=True, timeout=60)
    if expect_fail:
        assert p.returncode != 0
        return p.stderr
    assert p.returncode == 0, p.stderr
    return json.load(open(out))


def flatten(d):
    """meshes -> (world vertices per triangle corner, material per triangle, light record per triangle)"""
    P, mat, lights = [], [], []
    for m in d["meshes"]:
        v = np.array(m["P"], dtype=np.float32).reshape(-1, 3)
        idx = np.array(m["indices"]).reshape(-1, 3)
        for tri in idx:
            P.append(v[tri])
            mat.append(m["material"])
            lights.append((m["Le"], m["twosided"], m["nsamples"]) if m["emissive"] else None)
    return np.array(P, dtype=np.float32), mat, lights


@pytest.mark.parametrize("variant", ["diffuse", "microfacet", "specular", "uber", "mix"])
def test_cornell_scene_description_round_trip(tmp_path, variant):
    sc = scenes.cornell_box(variant)
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 96, 64, 16)
    d = dump(tmp_path, text)
    P, mat, lights = flatten(d)
    assert np.array_equal(P, sc.vertices[sc.indices])                 # world-space triangles, file order
    assert mat == sc.tri_material.tolist()
    for t, l in enumerate(lights):
        if sc.tri_light[t] < 0:
            assert l is None
        else:
            ref = sc.lights[sc.tri_light[t]]
            assert np.allclose(l[0], list(ref.L)) and l[1] == ref.two_sided and l[2] == max(1, ref.n_samples)
    for got, ref in zip(d["materials"], sc.materials[: sc.n_materials]):
        assert got["type"] == ref.type
        for k in ("kd", "ks", "kr", "kt", "eta", "k"):
            if (ref.type, k) in ((abi.MAT_MATTE, "kd"), (abi.MAT_METAL, "eta"), (abi.MAT_METAL, "k"), (abi.MAT_GLASS, "kr"), (abi.MAT_GLASS, "kt"),
                                 (abi.MAT_PLASTIC, "kd"), (abi.MAT_PLASTIC, "ks"), (abi.MAT_MIRROR, "kr"), (abi.MAT_TRANSLUCENT, "kd"),
                                 (abi.MAT_TRANSLUCENT, "ks"), (abi.MAT_SUBSTRATE, "kd"), (abi.MAT_SUBSTRATE, "ks"), (abi.MAT_UBER, "kd"),
                                 (abi.MAT_UBER, "ks"), (abi.MAT_UBER, "kr"), (abi.MAT_UBER, "kt")):
                assert np.array_equal(np.float32(got[k]), np.float32(list(getattr(ref, k))))
        if ref.type not in (abi.MAT_MATTE, abi.MAT_MIRROR, abi.MAT_MIX):
            assert np.float32(got["uroughness"]) == np.float32(ref.uroughness) and np.float32(got["vroughness"]) == np.float32(ref.vroughness)
            assert got["remap"] == ref.remap_roughness
        if ref.type == abi.MAT_TRANSLUCENT:
            assert np.array_equal(np.float32(got["reflect"]), np.float32(list(ref.reflect))) and np.array_equal(np.float32(got["transmit"]), np.float32(list(ref.transmit)))
        if ref.type == abi.MAT_UBER:
            assert np.array_equal(np.float32(got["opacity"]), np.float32(list(ref.opacity))) and np.float32(got["index"]) == np.float32(ref.index)
        if ref.type == abi.MAT_MIX:
            assert got["mix"] == [ref.mix_a, ref.mix_b] and np.array_equal(np.float32(got["amount"]), np.float32(list(ref.amount)))
    # camera / film / sampler / integrator agree with the Python builders the parity tests use
    cam, film = scenes.cornell_camera(96, 64), scenes.film_desc(96, 64)
    assert np.allclose(d["camera"]["c2w"], list(cam.camera_to_world), rtol=1e-6, atol=1e-4)
    assert np.allclose(d["camera"]["r2c"], list(cam.raster_to_camera), rtol=1e-6, atol=1e-9)
    assert d["film"]["cropped"] == list(film.cropped_pixel_bounds) and d["film"]["filter"] == "box"
    assert np.array_equal(np.float32(d["film"]["filter_table"]), np.float32(list(film.filter_table)))
    assert d["sampler"]["spp"] == 16 and d["sampler"]["sample_bounds"] == scenes.sample_bounds(film)
    assert d["integrator"] == {"name": "path", "maxdepth": 5, "rrthreshold": 1, "lightsamplestrategy": "power", "strategy": 0,
                               "pixel_bounds": scenes.sample_bounds(film)}


def test_many_lights_keeps_light_order_with_point_lights(tmp_path):
    sc = scenes.many_lights()
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4)
    d = dump(tmp_path, text)
    P, mat, lights = flatten(d)
    assert np.array_equal(P, sc.vertices[sc.indices])
    n_area = sum(1 for l in lights if l is not None)
    assert n_area + len(d["lights"]) == sc.n_lights
    pts = [sc.lights[i] for i in range(sc.n_lights) if sc.lights[i].type == abi.LIGHT_POINT]
    for got, ref in zip(d["lights"], pts):
        assert np.array_equal(np.float32(got["from"]), np.float32(list(ref.pos))) and np.array_equal(np.float32(got["I"]), np.float32(list(ref.L)))


def test_transforms_compose_as_the_format_defines(tmp_path):
    """CTM = CTM * T: the LAST transform written applies to the points FIRST; AttributeEnd restores; a mirroring
    scale flips the orientation flag; point lights sit at Translate(from) * CTM applied to the origin."""
    text = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 60
Film "image" "integer xresolution" 8 "integer yresolution" 8
Sampler "sobol" "integer pixelsamples" 1
Integrator "directlighting" "integer maxdepth" 3 "string strategy" "one"
WorldBegin
AttributeBegin
  Translate 10 0 0
  Rotate 90 0 0 1
  Scale 2 2 2
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 0 0  0 1 0  0 0 1]
AttributeEnd
AttributeBegin
  Scale -1 1 1
  Shape "trianglemesh" "point P" [1 0 0  0 1 0  0 0 1]
AttributeEnd
TransformBegin
  Translate 1 2 3
  LightSource "point" "rgb I" [2 3 4] "rgb scale" [2 2 2] "point from" [10 20 30]
TransformEnd
ConcatTransform [1 0 0 0  0 1 0 0  0 0 1 0  5 6 7 1]
Shape "trianglemesh" "point P" [0 0 0  1 0 0  0 1 0]
WorldEnd
'''
    d = dump(tmp_path, text)
    m0, m1, m2 = d["meshes"]
    # scale by 2, rotate 90 degrees about z ((x, y) -> (-y, x)), then translate by 10 in x
    assert np.allclose(np.array(m0["P"]).reshape(3, 3), [[10, 2, 0], [8, 0, 0], [10, 0, 2]], atol=1e-5)
    assert m0["reverse"] == 0 and m1["reverse"] == 1
    assert np.allclose(np.array(m1["P"]).reshape(3, 3), [[-1, 0, 0], [0, 1, 0], [0, 0, 1]])
    assert np.allclose(np.array(m2["P"]).reshape(3, 3), [[5, 6, 7], [6, 6, 7], [5, 7, 7]])          # column-major matrix: translation
    (l,) = d["lights"]
    assert np.allclose(l["from"], [11, 22, 33]) and np.allclose(l["I"], [4, 6, 8])
    assert d["integrator"]["name"] == "directlighting" and d["integrator"]["strategy"] == 1 and d["integrator"]["maxdepth"] == 3
    assert len(d["materials"]) == 1 and d["materials"][0]["type"] == abi.MAT_MATTE and np.allclose(d["materials"][0]["kd"], 0.5)   # default material
    # camera looking down +z from z = -5: camera-to-world is a translation
    c2w = np.array(d["camera"]["c2w"]).reshape(4, 4)
    assert np.allclose(c2w, [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -5], [0, 0, 0, 1]], atol=1e-6)


def test_defaults_crop_window_filters_and_pixel_bounds(tmp_path):
    text = '''
Camera "perspective"
PixelFilter "gaussian" "float xwidth" 1.5 "float ywidth" 1.5
Film "image" "float cropwindow" [0.25 0.75 0.25 0.5]
Sampler "sobol"
Integrator "path" "integer pixelbounds" [330 600 185 300] "string lightsamplestrategy" "uniform" "float rrthreshold" 0.5
WorldBegin
Material "plastic"
AreaLightSource "diffuse" "rgb L" [1 2 3] "rgb scale" [2 2 2] "integer nsamples" 4
Shape "trianglemesh" "point P" [0 0 5  1 0 5  0 1 5] "float uv" [0 0 1 0 0 1]
WorldEnd
'''
    d = dump(tmp_path, text)
    assert (d["film"]["xres"], d["film"]["yres"]) == (1280, 720) and d["film"]["cropped"] == [320, 180, 960, 360]
    film = scenes.film_desc(1280, 720, "gaussian", (1.5, 1.5), crop=(0.25, 0.75, 0.25, 0.5))
    assert np.allclose(d["film"]["filter_table"], list(film.filter_table), rtol=1e-6, atol=1e-9)
    sb = scenes.sample_bounds(film)
    assert d["sampler"] == {"name": "sobol", "samplepixelcenter": False, "spp": 16, "sample_bounds": sb}
    assert d["integrator"]["pixel_bounds"] == [max(sb[0], 330), max(sb[1], 185), min(sb[2], 600), min(sb[3], 300)]
    assert d["integrator"]["lightsamplestrategy"] == "uniform" and d["integrator"]["rrthreshold"] == 0.5
    m = d["materials"][0]
    assert m["type"] == abi.MAT_PLASTIC and np.allclose(m["kd"], 0.25) and np.allclose(m["ks"], 0.25) and np.float32(m["uroughness"]) == np.float32(0.1) and m["remap"] == 1
    mesh = d["meshes"][0]
    assert mesh["emissive"] == 1 and np.allclose(mesh["Le"], [2, 4, 6]) and mesh["nsamples"] == 4 and mesh["twosided"] == 0
    assert d["camera"]["focaldistance"][0] >= 1e30 and d["camera"]["lensradius"] == [0]
    # fov 90 on the shorter (y) axis of a 16:9 frame: raster (640, 0) maps to camera (0, 1, 1) up to scale
    r2c = np.array(d["camera"]["r2c"]).reshape(4, 4)
    p = r2c @ np.array([640.0, 0.0, 0.0, 1.0])
    p = p[:3] / p[3]
    assert np.allclose(p[:2] / p[2], [0.0, 1.0], atol=1e-5)
    assert not any("uv" in w for w in d["warnings"])


@pytest.mark.parametrize("text,needle", [
    ('WorldBegin\nShape "cylinder" "float radius" 1\nWorldEnd\n', "cylinder"),
    ('WorldBegin\nTexture "t" "spectrum" "imagemap" "string filename" "x.png"\nWorldEnd\n', "PFM"),
    ('WorldBegin\nMaterial "matte" "texture Kd" "t"\nWorldEnd\n', "texture"),
    ('WorldBegin\nTexture "f" "float" "checkerboard"\nWorldEnd\n', "float class"),
    ('WorldBegin\nMaterial "plastic" "texture roughness" "nope"\nWorldEnd\n', "only constant values"),
    ('WorldBegin\nTexture "t" "spectrum" "constant"\nMaterial "uber" "texture opacity" "t"\nWorldEnd\n', "constant values only"),
    ('WorldBegin\nMaterial "subsurface"\nWorldEnd\n', "subsurface"),
    ('WorldBegin\nMakeNamedMaterial "a" "string type" "none"\nMaterial "mix" "string namedmaterial1" "a" "string namedmaterial2" "a"\nWorldEnd\n', "two materials"),
    ('WorldBegin\nShape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0] "vector S" [0 0 1 0 0 1 0 0 1]\nWorldEnd\n', "tangents"),
    ('Sampler "stratified"\nWorldBegin\nWorldEnd\n', "sobol and halton"),
    ('Sampler "random"\nWorldBegin\nWorldEnd\n', "generator that runs on"),
    ('Integrator "bdpt"\nSampler "sobol"\nWorldBegin\nWorldEnd\n', "bdpt"),
    ('WorldBegin\nCamera "perspective"\nWorldEnd\n', "world block"),
    ('Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\n', "world block"),
    ('WorldBegin\nMaterial "matte" "spectrum Kd" "metals/Cu.spd"\nWorldEnd\n', "rgb"),
    ('WorldBegin\nShape "trianglemesh" "integer indices" [0 1 5] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n', "out-of-bounds"),
    ('WorldBegin\nFrobnicate 1 2 3\nWorldEnd\n', "unknown directive"),
    ('WorldBegin\nMaterial "matte" "rgb Kd" [0.5 0.5\nWorldEnd\n', "WorldEnd"),
])
def test_front_end_refuses_what_it_cannot_render(tmp_path, text, needle):
    err = dump(tmp_path, text, expect_fail=True)
    assert needle in err, err


def test_metal_without_eta_and_k_is_the_reference_copper(tmp_path):
    """CreateMetalMaterial's defaults (Enigma/Material.cpp:604-613): copper's spectra as RGB, roughness 0.5, remapped."""
    import raytrace_amd as ra
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nMaterial "metal"\nShape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\n'
                       'Material "metal" "rgb k" [1 2 3]\nShape "trianglemesh" "point P" [0 0 1 1 0 1 0 1 1]\nWorldEnd\n')
    eta, k = ra.default_metal_spectra()
    m0, m1 = d["materials"][0], d["materials"][1]
    assert m0["type"] == abi.MAT_METAL and m0["remap"] == 1 and m0["uroughness"] == 0.5 and m0["vroughness"] == 0.5
    assert np.allclose(m0["eta"], eta, rtol=1e-6) and np.allclose(m0["k"], k, rtol=1e-6)
    assert np.allclose(m1["eta"], eta, rtol=1e-6) and m1["k"] == [1, 2, 3]        # each default on its own


def test_defaults_of_the_lobe_list_materials_and_the_mix_of_an_unknown_name(tmp_path):
    """Create{Mirror,Translucent,Substrate,Uber,Mix}Material's defaults (Enigma/Material.cpp:617-621,648-653,673-682,720-748);
    an uber's roughnesses as its constructor resolves them (:563-570); a mix of a name nobody made takes a matte built from the
    mix's own parameters (Enigma/Api.cpp:157-173); an unknown material is a matte, with a warning (:189-192)."""
    tri = 'Shape "trianglemesh" "point P" [0 0 %d 1 0 %d 0 1 %d]\n'
    text = 'Sampler "sobol"\nWorldBegin\n'
    for k, m in enumerate(['Material "mirror"', 'Material "translucent"', 'Material "substrate"', 'Material "uber"',
                           'Material "uber" "float roughness" 0.3 "float vroughness" 0.2 "float eta" 1.2 "float index" 1.9 "rgb opacity" [0.5 0.5 0.5]',
                           'Material "uber" "float uroughness" 0.4',
                           'MakeNamedMaterial "a" "string type" "mirror"\nMaterial "mix" "string namedmaterial1" "a" "string namedmaterial2" "nobody" "rgb Kd" [0.1 0.2 0.3]',
                           'Material "frobnicate" "rgb Kd" [0.3 0.2 0.1]']):
        text += m + "\n" + tri % (k, k, k)
    d = dump(tmp_path, text + "WorldEnd\n")
    ms = d["materials"]
    f32 = np.float32
    assert ms[0]["type"] == abi.MAT_MIRROR and np.allclose(ms[0]["kr"], 0.9)
    t = ms[1]
    assert t["type"] == abi.MAT_TRANSLUCENT and np.allclose(t["kd"], 0.25) and np.allclose(t["ks"], 0.25) and np.allclose(t["reflect"], 0.5) and np.allclose(t["transmit"], 0.5)
    assert f32(t["uroughness"]) == f32(0.1) and t["remap"] == 1
    sub = ms[2]
    assert sub["type"] == abi.MAT_SUBSTRATE and np.allclose(sub["kd"], 0.5) and np.allclose(sub["ks"], 0.5) and f32(sub["uroughness"]) == f32(0.1) and f32(sub["vroughness"]) == f32(0.1)
    u = ms[3]
    assert u["type"] == abi.MAT_UBER and np.allclose(u["kd"], 0.25) and np.allclose(u["ks"], 0.25) and np.allclose(u["kr"], 0) and np.allclose(u["kt"], 0)
    assert np.allclose(u["opacity"], 1) and f32(u["uroughness"]) == f32(0.1) and f32(u["vroughness"]) == f32(0.1) and f32(u["index"]) == f32(1.5)
    assert f32(ms[4]["uroughness"]) == f32(0.3) and f32(ms[4]["vroughness"]) == f32(0.2) and f32(ms[4]["index"]) == f32(1.2) and np.allclose(ms[4]["opacity"], 0.5)
    assert f32(ms[5]["uroughness"]) == f32(0.4) and f32(ms[5]["vroughness"]) == f32(0.4)
    # materials in creation order: ..., the named mirror, the stand-in matte, the mix; then the unknown one as a matte
    mirror, standin, mix, unknown = ms[6], ms[7], ms[8], ms[9]
    assert mirror["type"] == abi.MAT_MIRROR and standin["type"] == abi.MAT_MATTE and np.allclose(standin["kd"], [0.1, 0.2, 0.3])
    assert mix["type"] == abi.MAT_MIX and mix["mix"] == [6, 7] and np.allclose(mix["amount"], 0.5)
    assert unknown["type"] == abi.MAT_MATTE and np.allclose(unknown["kd"], [0.3, 0.2, 0.1])
    assert [m["material"] for m in d["meshes"]] == [0, 1, 2, 3, 4, 5, 8, 9]
    assert any("nobody" in w for w in d["warnings"]) and any("frobnicate" in w for w in d["warnings"])


def test_plymesh_reads_what_the_references_rply_callbacks_keep(tmp_path):
    """Shape "plymesh" (CreatePLYMesh and its callbacks, Enigma/TriangleMesh.cpp:98-174,791-931): ascii and both binary
    byte orders; quads become (0 1 2) (3 0 2); faces of other sizes are skipped; normals and the first complete uv pair are
    kept; other elements and properties are read past; the mesh goes through the CTM like a trianglemesh."""
    rng = np.random.RandomState(5)
    v = rng.uniform(-1, 1, (7, 3)).astype(np.float32)
    nrm = rng.normal(size=(7, 3)).astype(np.float32)
    uv = rng.uniform(0, 1, (7, 2)).astype(np.float32)
    tris, quads = [[0, 1, 2], [2, 3, 4]], [[3, 4, 5, 6]]
    want_idx = [0, 1, 2, 2, 3, 4, 3, 4, 5, 6, 3, 5]
    for k, fmt in enumerate(("ascii", "binary_little_endian", "binary_big_endian")):
        pbrt_writer.write_ply(str(tmp_path / ("m%d.ply" % k)), v, tris, normals=nrm, uvs=uv, fmt=fmt, quads=quads)
    # a hand-written file: doubles, s / t names, a property and an element nobody reads, a pentagon, shorts as indices
    with open(tmp_path / "hand.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment by hand\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\n"
                "property uchar red\nproperty float s\nproperty float t\nelement edge 1\nproperty int a\nproperty int b\n"
                "element face 3\nproperty list uchar short vertex_indices\nproperty int face_indices\nend_header\n")
        for i in range(5):
            f.write("%d.25 %d %d 200 0.%d 0.5\n" % (i, i * 2, -i, i))
        f.write("7 9\n3 0 1 2 11\n5 0 1 2 3 4 12\n4 1 2 3 4 13\n")
    text = 'Sampler "sobol"\nWorldBegin\nTranslate 1 2 3\nScale 2 2 2\n'
    for k in range(3):
        text += 'Shape "plymesh" "string filename" "m%d.ply"\n' % k
    text += 'Shape "trianglemesh" "integer indices" [%s] "point P" [%s] "normal N" [%s] "float uv" [%s]\n' % (
        " ".join(map(str, want_idx)), " ".join("%.9g" % x for x in v.reshape(-1)), " ".join("%.9g" % x for x in nrm.reshape(-1)), " ".join("%.9g" % x for x in uv.reshape(-1)))
    text += 'Shape "plymesh" "string filename" "hand.ply"\nWorldEnd\n'
    d = dump(tmp_path, text)
    ref = d["meshes"][3]
    for k in range(3):
        m = d["meshes"][k]
        assert m["indices"] == want_idx == ref["indices"]
        for key in ("P", "N", "uv"):
            assert np.array_equal(np.float32(m[key]), np.float32(d["meshes"][0][key])), (k, key)      # the three encodings hold the same floats
            assert np.allclose(np.float32(m[key]), np.float32(ref[key]), rtol=1e-6, atol=1e-7), (k, key)   # the inline mesh goes through nine decimal digits
    assert np.allclose(np.float32(ref["P"]).reshape(-1, 3), v * 2 + np.array([1, 2, 3], dtype=np.float32), rtol=1e-6)
    hand = d["meshes"][4]
    assert hand["indices"] == [0, 1, 2, 1, 2, 3, 4, 1, 3]                 # the pentagon is skipped, the quad split
    P = np.float32(hand["P"]).reshape(-1, 3)
    assert np.allclose(P[1], np.array([1.25, 2, -1]) * 2 + [1, 2, 3]) and np.allclose(np.float32(hand["uv"]).reshape(-1, 2)[3], [0.3, 0.5])
    assert hand["N"] == [] and any("neither triangles nor quads" in w for w in d["warnings"])
    # errors: a missing file, an index out of range, a truncated binary file, a file without faces
    assert "Couldn't open PLY file" in dump(tmp_path, 'WorldBegin\nShape "plymesh" "string filename" "nope.ply"\nWorldEnd\n', expect_fail=True)
    pbrt_writer.write_ply(str(tmp_path / "bad.ply"), v, [[0, 1, 9]], fmt="ascii")
    assert "out of bounds" in dump(tmp_path, 'WorldBegin\nShape "plymesh" "string filename" "bad.ply"\nWorldEnd\n', expect_fail=True)
    data = open(tmp_path / "m1.ply", "rb").read()
    open(tmp_path / "cut.ply", "wb").write(data[:-7])
    assert "unable to read the contents" in dump(tmp_path, 'WorldBegin\nShape "plymesh" "string filename" "cut.ply"\nWorldEnd\n', expect_fail=True)
    with open(tmp_path / "nofaces.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n")
    assert "No face/vertex elements" in dump(tmp_path, 'WorldBegin\nShape "plymesh" "string filename" "nofaces.ply"\nWorldEnd\n', expect_fail=True)
    # the other common name of the face list is not the one the reference asks rply for: no shape, a warning
    with open(tmp_path / "other.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                "property list uchar int vertex_index\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nShape "plymesh" "string filename" "other.ply"\nWorldEnd\n')
    assert d["meshes"] == [] and any("vertex_indices" in w for w in d["warnings"])


def test_environment_camera_reaches_the_dump(tmp_path):
    d = dump(tmp_path, 'LookAt 1 2 3  1 2 10  0 1 0\nCamera "environment" "float lensradius" 3\nSampler "sobol"\nWorldBegin\nWorldEnd\n')
    assert d["camera"]["type"] == "environment"
    ref = scenes.environment_camera_desc((1, 2, 3), (1, 2, 10), (0, 1, 0))
    assert np.allclose(d["camera"]["c2w"], list(ref.camera_to_world), atol=1e-6)
    assert dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nWorldEnd\n')["camera"]["type"] == "perspective"
    assert "orthographic" in dump(tmp_path, 'Camera "orthographic"\nSampler "sobol"\nWorldBegin\nWorldEnd\n', expect_fail=True)


def test_shapes_whose_factories_are_empty_in_the_reference_make_nothing(tmp_path):
    """curve, loopsubdiv and nurbs return no shape in the reference (Enigma/Curve.cpp:10-14, Enigma/SubDiv.cpp:6-9,
    Enigma/Nurbs.cpp:10-14); an unknown shape name is a warning (Enigma/Api.cpp:123-124)."""
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nShape "loopsubdiv" "integer nlevels" 3\nShape "curve"\nShape "nurbs"\nShape "teapot"\n'
                       'Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    assert len(d["meshes"]) == 1 and d["spheres"] == []
    assert sum("no shape" in w for w in d["warnings"]) == 3 and any("teapot" in w and "unknown" in w for w in d["warnings"])


def test_include_comments_and_named_materials(tmp_path):
    with open(tmp_path / "geo.pbrt", "w") as f:
        f.write('# included file\nNamedMaterial "red"\nShape "trianglemesh" "point P" [0 0 0  1 0 0  0 1 0] # trailing comment\n')
    text = '''
Sampler "sobol" "integer pixelsamples" [ 3 ]   # rounded up to 4 by the Sobol sampler
WorldBegin
MakeNamedMaterial "red" "string type" "matte" "color Kd" [ .65 .05 .05 ]
MakeNamedMaterial "mirrorish" "string type" "glass" "rgb Kt" [0 0 0] "float eta" 1.33
Material "metal" "rgb eta" [0.2 0.9 1.1] "rgb k" [3.9 2.4 2.1] "float roughness" 0.2
AttributeBegin
  Include "geo.pbrt"
AttributeEnd
Shape "trianglemesh" "point P" [0 0 1  1 0 1  0 1 1]
NamedMaterial "mirrorish"
Shape "trianglemesh" "point P" [0 0 2  1 0 2  0 1 2]
WorldEnd
'''
    d = dump(tmp_path, text)
    assert d["sampler"]["spp"] == 4
    assert [m["material"] for m in d["meshes"]] == [0, 2, 1]
    red, glass, metal = d["materials"]
    assert np.allclose(red["kd"], [0.65, 0.05, 0.05]) and glass["type"] == abi.MAT_GLASS and np.float32(glass["index"]) == np.float32(1.33)
    assert np.allclose(glass["kt"], 0) and np.allclose(glass["kr"], 1)
    assert np.float32(metal["uroughness"]) == np.float32(0.2) and np.float32(metal["vroughness"]) == np.float32(0.2)


def test_spot_and_distant_lights(tmp_path):
    """CreateSpotLight / CreateDistantLight (Enigma/Lights.cpp:752-771,791-799): a spot light sits at CTM(from) and its
    world-to-light rotation takes the direction from -> to to +z; a distant light keeps CTM(from - to) unnormalised."""
    text = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 60
Film "image" "integer xresolution" 8 "integer yresolution" 8
Sampler "sobol" "integer pixelsamples" 1
WorldBegin
TransformBegin
  Translate 1 2 3
  Scale 2 2 2
  LightSource "spot" "rgb I" [2 3 4] "point from" [0 5 0] "point to" [0 0 0] "float coneangle" 40 "float conedeltaangle" 15
  LightSource "distant" "rgb L" [1 2 3] "rgb scale" [2 2 2] "point from" [0 3 0] "point to" [0 0 4]
TransformEnd
LightSource "spot" "point to" [3 0 4]
AttributeBegin
  Rotate 90 1 0 0
  LightSource "infinite" "rgb L" [1 2 3] "rgb scale" [0.5 0.5 0.5] "integer samples" 4
AttributeEnd
Shape "trianglemesh" "point P" [0 0 0  1 0 0  0 1 0]
WorldEnd
'''
    d = dump(tmp_path, text)
    spot, dist, spot2, inf = d["lights"]
    assert inf["type"] == abi.LIGHT_INFINITE and np.allclose(inf["I"], [0.5, 1.0, 1.5]) and inf["nsamples"] == 4
    l2w, w2l = np.array(inf["light_to_world"]).reshape(3, 3), np.array(inf["world_to_light"]).reshape(3, 3)
    assert np.allclose(l2w @ [0, 0, 1], [0, -1, 0], atol=1e-6) and np.allclose(l2w @ w2l, np.eye(3), atol=1e-6)   # Rotate 90 about x
    assert spot["type"] == abi.LIGHT_SPOT and dist["type"] == abi.LIGHT_DISTANT
    assert np.allclose(spot["from"], [1, 12, 3]) and np.allclose(spot["I"], [2, 3, 4])
    assert np.isclose(spot["cos_total_width"], np.cos(np.radians(40.0)), atol=1e-6) and np.isclose(spot["cos_falloff_start"], np.cos(np.radians(25.0)), atol=1e-6)
    w2l = np.array(spot["world_to_light"]).reshape(3, 3)
    down = w2l @ np.array([0.0, -1.0, 0.0])                 # world direction the light looks along
    assert np.allclose(down / np.linalg.norm(down), [0, 0, 1], atol=1e-6)
    assert np.isclose(np.linalg.norm(down), 0.5, atol=1e-6)  # the CTM's scale 2 is inverted, as m_worldToLight would be
    assert np.allclose(dist["dir"], [0, 6, -8]) and np.allclose(dist["I"], [2, 4, 6])
    # defaults: from the origin towards `to`, 30 degree cone with a 5 degree falloff band, I = 1
    assert np.allclose(spot2["from"], [0, 0, 0]) and np.allclose(spot2["I"], [1, 1, 1])
    assert np.isclose(spot2["cos_total_width"], np.cos(np.radians(30.0)), atol=1e-6) and np.isclose(spot2["cos_falloff_start"], np.cos(np.radians(25.0)), atol=1e-6)
    assert np.allclose(np.array(spot2["world_to_light"]).reshape(3, 3) @ np.array([0.6, 0.0, 0.8]), [0, 0, 1], atol=1e-6)


def test_media_medium_interfaces_and_the_volpath_integrator(tmp_path):
    """MakeNamedMedium / MediumInterface / Material "none" / Integrator "volpath" (Enigma/Api.cpp:135,282-331,644-662,
    1017-1036; Enigma/VolPathIntegrator.cpp:179-200) against the Python-built fog scene."""
    sc = scenes.cornell_fog("microfacet")
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4,
                                  integrator='Integrator "volpath" "integer maxdepth" 7 "float rrthreshold" 0.5 "string lightsamplestrategy" "uniform"')
    d = dump(tmp_path, text)
    assert d["integrator"]["name"] == "volpath" and d["integrator"]["maxdepth"] == 7 and d["integrator"]["rrthreshold"] == 0.5
    assert d["integrator"]["lightsamplestrategy"] == "uniform"
    assert len(d["media"]) == 2
    for got, ref in zip(d["media"], [sc.media[0], sc.media[1]]):
        assert np.array_equal(np.float32(got["sigma_a"]), np.float32(list(ref.sigma_a))) and np.array_equal(np.float32(got["sigma_s"]), np.float32(list(ref.sigma_s)))
        assert np.float32(got["g"][0]) == np.float32(ref.g)
    tri_mat, tri_in, tri_out = [], [], []
    for m in d["meshes"]:
        n = len(m["indices"]) // 3
        tri_mat += [m["material"]] * n; tri_in += [m["medium_inside"]] * n; tri_out += [m["medium_outside"]] * n
    assert tri_in == list(sc.tri_medium_inside) and tri_out == list(sc.tri_medium_outside)
    assert [(-1 if int(x) == abi.NO_MATERIAL else int(x)) for x in sc.tri_material] == tri_mat
    # a medium named before it is made, a heterogeneous medium without its grid or with a grid of the wrong size: reported, not guessed
    head = 'Camera "perspective"\nFilm "image" "integer xresolution" 4 "integer yresolution" 4\nSampler "sobol"\nWorldBegin\n'
    for body, msg in (('MediumInterface "fog" ""\nShape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n', "undefined"),
                      ('MakeNamedMedium "a" "string type" "heterogeneous"\nWorldEnd\n', "No \"density\" values"),
                      ('MakeNamedMedium "a" "string type" "heterogeneous" "float density" [1 2 3] "integer nx" 2\nWorldEnd\n', "expected nx*ny*nz = 2"),
                      ('MakeNamedMedium "a" "string type" "foam"\nWorldEnd\n', "unknown")):
        f = tmp_path / "bad.pbrt"
        f.write_text(head + body)
        p = subprocess.run([EXE, "--dump", str(f), str(tmp_path / "bad.json")], capture_output=True, text=True)
        assert p.returncode != 0 and msg in p.stderr


def test_heterogeneous_medium_is_the_references_grid_density_medium(tmp_path):
    """MakeNamedMedium "heterogeneous" (Enigma/Api.cpp:303-326) makes the reference's GridDensityMedium, whose Tr and Sample
    return black (Enigma/Medium.cpp:51-59): the front end runs the factory's checks and hands the medium on as that kind."""
    f = tmp_path / "grid.pbrt"
    f.write_text('Camera "perspective"\nFilm "image" "integer xresolution" 4 "integer yresolution" 4\nSampler "sobol"\nWorldBegin\n'
                 'MakeNamedMedium "smoke" "string type" "heterogeneous" "rgb sigma_a" [1 1 1] "rgb sigma_s" [2 2 2] "float g" 0.3 "float scale" 2 '
                 '"integer nx" 2 "integer ny" 2 "integer nz" 1 "float density" [0.1 0.2 0.3 0.4] "point p0" [0 0 0] "point p1" [1 2 3]\n'
                 'MakeNamedMedium "fog" "string type" "homogeneous"\n'
                 'MediumInterface "smoke" ""\nMaterial ""\nShape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0] "integer indices" [0 1 2]\nWorldEnd\n')
    out = tmp_path / "grid.json"
    p = subprocess.run([EXE, "--dump", str(f), str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    media = json.loads(out.read_text())["media"]
    assert [m["kind"] for m in media] == [abi.MEDIUM_GRID_DENSITY, abi.MEDIUM_HOMOGENEOUS]
    assert media[0]["sigma_a"] == [2.0, 2.0, 2.0] and media[0]["sigma_s"] == [4.0, 4.0, 4.0] and abs(media[0]["g"][0] - 0.3) < 1e-7


def _presets():
    rows = [l.rstrip("\n").split(" | ") for l in open(os.path.join(ROOT, "raytrace_amd", "data", "medium_presets.txt")) if not l.startswith("#")]
    return [r[0] for r in rows], np.array([[float(t) for t in (r[1] + " " + r[2]).split()] for r in rows], dtype="<f4")


def test_measured_medium_presets_are_the_reference_table(tmp_path):
    """MakeNamedMedium's "preset" (Enigma/Api.cpp:285-298, GetMediumScatteringProperties, Enigma/MediumInterface.cpp:184-194):
    the table of measured media is pinned to the reference's by digest (tests/golden/medium_presets_digest.json, written
    by tools/derive_medium_presets.py from the reference's text); a preset sets sigma_a and sigma'_s, explicit spectra
    override it, "scale" multiplies either, and an unknown name warns and keeps the defaults (those of "Wholemilk")."""
    import hashlib
    names, coeff = _presets()
    digest = json.load(open(os.path.join(ROOT, "tests", "golden", "medium_presets_digest.json")))
    assert len(names) == digest["count"] == 47
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == digest["names"]
    assert hashlib.sha256(coeff.tobytes()).hexdigest() == digest["coefficients"]
    head = 'Camera "perspective"\nFilm "image" "integer xresolution" 4 "integer yresolution" 4\nSampler "sobol"\nWorldBegin\n'
    body = "".join('MakeNamedMedium "m%d" "string type" "homogeneous" "string preset" "%s"\n' % (i, n) for i, n in enumerate(names))
    body += 'MakeNamedMedium "scaled" "string type" "homogeneous" "string preset" "Skin1" "float scale" 4 "float g" 0.5\n'
    body += 'MakeNamedMedium "mixed" "string type" "homogeneous" "string preset" "Ketchup" "rgb sigma_s" [1 2 3]\n'
    body += 'MakeNamedMedium "unknown" "string type" "homogeneous" "string preset" "Unobtainium"\n'
    body += 'MakeNamedMedium "plain" "string type" "homogeneous"\n'
    d = dump(tmp_path, head + body + 'Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    media = d["media"]
    assert len(media) == len(names) + 4
    for i in range(len(names)):  # the compiled table (what the driver was built with) against the data file, value for value
        assert np.array_equal(np.float32(media[i]["sigma_s"]), coeff[i, :3]) and np.array_equal(np.float32(media[i]["sigma_a"]), coeff[i, 3:]), names[i]
        assert media[i]["g"][0] == 0.0
    skin, ketchup, milk = names.index("Skin1"), names.index("Ketchup"), names.index("Wholemilk")
    scaled, mixed, unknown, plain = media[len(names):]
    assert np.array_equal(np.float32(scaled["sigma_a"]), coeff[skin, 3:] * np.float32(4)) and np.array_equal(np.float32(scaled["sigma_s"]), coeff[skin, :3] * np.float32(4))
    assert scaled["g"][0] == 0.5
    assert np.array_equal(np.float32(mixed["sigma_a"]), coeff[ketchup, 3:]) and np.array_equal(np.float32(mixed["sigma_s"]), np.float32([1, 2, 3]))
    for m in (unknown, plain):
        assert np.array_equal(np.float32(m["sigma_a"]), coeff[milk, 3:]) and np.array_equal(np.float32(m["sigma_s"]), coeff[milk, :3])
    assert any("Unobtainium" in w for w in d["warnings"])


def test_shading_normals_and_uvs_reach_the_mesh(tmp_path):
    """"normal N" goes through the inverse transpose of the CTM (Enigma/TriangleMesh.cpp:44-48, Enigma/Transform.h:54-59),
    uvs are taken from "point2 uv" / "float uv" / "st" (Enigma/TriangleMesh.cpp:693-708) and dropped when too few,
    ReverseOrientation and a mirroring CTM are kept apart (Enigma/TriangleMesh.cpp:288,369)."""
    sc = scenes.cornell_smooth()
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4)
    d = dump(tmp_path, text)
    N, UV, rev = [], [], []
    for m in d["meshes"]:
        idx = np.array(m["indices"]).reshape(-1, 3)
        n = np.array(m["N"], dtype=np.float32).reshape(-1, 3) if m["N"] else None
        uv = np.array(m["uv"], dtype=np.float32).reshape(-1, 2) if m["uv"] else None
        for tri in idx:
            N.append(None if n is None else n[tri])
            UV.append(None if uv is None else uv[tri])
            rev.append((m["reverse"], m["reverse_orientation"]))
    for t in range(sc.indices.shape[0]):
        fl = int(sc.tri_flags[t])
        assert (N[t] is not None) == bool(fl & abi.TRI_HAS_NORMALS) and (UV[t] is not None) == bool(fl & abi.TRI_HAS_UVS)
        if N[t] is not None:
            assert np.allclose(N[t], sc.normals[sc.indices[t]], rtol=1e-6, atol=1e-7)
        if UV[t] is not None:
            assert np.array_equal(UV[t], sc.uvs[sc.indices[t]])
        assert rev[t] == (fl & 1, 1 if fl & abi.TRI_REVERSE_ORIENTATION else 0)
    head = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 60
Film "image" "integer xresolution" 8 "integer yresolution" 8
Sampler "sobol" "integer pixelsamples" 1
Integrator "path"
WorldBegin
'''
    d = dump(tmp_path, head + '''
AttributeBegin
  Scale 2 4 -1
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 0 0  0 1 0  0 0 1] "normal N" [1 0 0  0 1 0  1 1 1] "point2 st" [0 0  1 0  0 1]
AttributeEnd
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 0 0  0 1 0  0 0 1] "float uv" [0 0 1 0]
WorldEnd
''')
    m0, m1 = d["meshes"]
    assert np.allclose(np.array(m0["N"]).reshape(3, 3), [[0.5, 0, 0], [0, 0.25, 0], [0.5, 0.25, -1]])   # n * M^-1
    assert np.allclose(np.array(m0["uv"]).reshape(3, 2), [[0, 0], [1, 0], [0, 1]])
    assert m0["reverse"] == 1 and m0["reverse_orientation"] == 0                                          # the mirror flips ng only
    assert m1["uv"] == [] and m1["N"] == []                                                                # too few uvs: discarded
    err = dump(tmp_path, head + 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 0 0  0 1 0  0 0 1] "vector S" [1 0 0 1 0 0 1 0 0]\nWorldEnd\n', expect_fail=True)
    assert "tangents" in err


def test_image_textures_reach_the_scene(tmp_path):
    """Texture "<name>" "spectrum" "imagemap" (CreateImageSpectrumTexture, Enigma/Texture.cpp:412-453) read from PFM files and
    bound through "texture Kd" (TextureParams::GetSpectrumTexture, Enigma/ParameterSet.cpp:686-743); "constant" textures
    become the constant; gamma defaults to false for a PFM file (it is true for .png / .tga only)."""
    sc = scenes.cornell_textured(True)
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4, texture_dir=str(tmp_path))
    d = dump(tmp_path, text)
    assert len(d["textures"]) == sc.n_textures
    for got, ref in zip(d["textures"], sc.textures[: sc.n_textures]):
        rgb = np.ctypeslib.as_array(__import__("ctypes").cast(ref.rgb, __import__("ctypes").POINTER(__import__("ctypes").c_float)), shape=(ref.height, ref.width, 3))
        assert (got["width"], got["height"], got["trilinear"], got["wrap"]) == (ref.width, ref.height, ref.trilinear, ref.wrap)
        assert got["gamma"] == ref.gamma and np.float32(got["scale"]) == np.float32(ref.scale)
        assert np.allclose(got["mapping"], [ref.su, ref.sv, ref.du, ref.dv])
        assert np.allclose(got["first_texel"], rgb[0, 0], rtol=1e-6) and np.isclose(got["sum"], float(rgb.astype(np.float64).sum()), rtol=1e-5)
    for got, ref in zip(d["materials"], sc.materials[: sc.n_materials]):
        assert got["tex"] == [ref.tex_kd, ref.tex_ks, ref.tex_kr, ref.tex_kt, ref.tex_eta, ref.tex_k]
    P, mat, lights = flatten(d)
    assert np.array_equal(P, sc.vertices[sc.indices])
    pbrt_writer.write_pfm(str(tmp_path / "one.pfm"), np.full((2, 2, 3), 0.25, np.float32))
    head = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 60
Film "image" "integer xresolution" 8 "integer yresolution" 8
Sampler "sobol" "integer pixelsamples" 1
Integrator "path"
WorldBegin
'''
    d = dump(tmp_path, head + '''
Texture "c" "color" "constant" "rgb value" [0.1 0.2 0.3]
Texture "img" "spectrum" "imagemap" "string filename" "one.pfm"
Material "plastic" "texture Kd" "c" "texture Ks" "img"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 0 0  0 1 0  0 0 1]
WorldEnd
''')
    m = d["materials"][-1]
    assert np.allclose(m["kd"], [0.1, 0.2, 0.3]) and m["tex"] == [0, 1, 0, 0, 0, 0]
    t = d["textures"][0]
    assert (t["trilinear"], t["gamma"], t["wrap"]) == (0, 0, 0)
    for body, needle in (('Texture "f" "float" "windy"', "float class"),
                         ('Texture "x" "spectrum" "checkerboard"', "not supported"),
                         ('Texture "x" "spectrum" "imagemap" "string filename" "one.png"', "PFM"),
                         ('Texture "x" "spectrum" "imagemap" "string filename" "missing.pfm"', "unable to read"),
                         ('Material "matte" "texture Kd" "nope"', "Couldn\'t find")):
        err = dump(tmp_path, head + body + "\nWorldEnd\n", expect_fail=True)
        assert needle in err, err


def test_sphere_shapes_reach_the_scene(tmp_path):
    """Shape "sphere" (CreateSphereShape, Enigma/ShapeImp.cpp:723-732): both matrices of the CTM, the constructor's
    parameters with their defaults, the orientation flag = ReverseOrientation ^ a mirroring CTM, the bound material."""
    sc = scenes.cornell_spheres(lamp=True)
    text = pbrt_writer.scene_text(sc, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4, texture_dir=str(tmp_path))
    d = dump(tmp_path, text)
    assert len(d["spheres"]) == sc.n_spheres
    (lamp,) = d["lights"]             # AreaLightSource in front of Shape "sphere": a DiffuseAreaLight on the last sphere, after the mesh lights
    ref = sc.lights[sc.n_lights - 1]
    assert (lamp["type"], lamp["sphere"], lamp["nsamples"], lamp["twosided"]) == (abi.LIGHT_DIFFUSE_AREA_SPHERE, sc.n_spheres - 1, 2, 0) and np.allclose(lamp["I"], list(ref.L))
    assert lamp["order"] == len(d["meshes"])
    for got, ref in zip(d["spheres"], sc.spheres[: sc.n_spheres]):
        assert np.allclose(got["o2w"], list(ref.object_to_world), rtol=1e-6, atol=1e-6)
        assert np.allclose(got["w2o"], list(ref.world_to_object), rtol=1e-5, atol=1e-6)
        assert np.allclose([got["radius"], got["zmin"], got["zmax"], got["phimax"]], [ref.radius, ref.z_min, ref.z_max, ref.phi_max])
        assert got["material"] == ref.material and got["flags"] == ref.flags
    fog = scenes.cornell_spheres(fog=True)
    d = dump(tmp_path, pbrt_writer.scene_text(fog, (278, 273, -800), (278, 273, 0), (0, 1, 0), FOV, 32, 32, 4, texture_dir=str(tmp_path),
                                               integrator='Integrator "volpath" "integer maxdepth" 5 "string lightsamplestrategy" "power"'))
    assert [(s["medium_inside"], s["medium_outside"]) for s in d["spheres"]] == [(s.medium_inside - 1, s.medium_outside - 1) for s in fog.spheres[: fog.n_spheres]]
    assert any(s["medium_inside"] >= 0 for s in d["spheres"])
    head = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 60
Film "image" "integer xresolution" 8 "integer yresolution" 8
Sampler "sobol" "integer pixelsamples" 1
Integrator "path"
WorldBegin
'''
    d = dump(tmp_path, head + '''
AttributeBegin
  Translate 1 2 3
  Scale 2 2 -2
  Shape "sphere" "float radius" 3
AttributeEnd
WorldEnd
''')
    s0 = d["spheres"][0]
    assert (s0["radius"], s0["zmin"], s0["zmax"], s0["phimax"], s0["flags"]) == (3, -3, 3, 360, 1)           # the mirror alone flips the normal
    assert np.allclose(np.array(s0["o2w"]).reshape(4, 4), [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, -2, 3], [0, 0, 0, 1]])
    assert np.allclose(np.array(s0["o2w"]).reshape(4, 4) @ np.array(s0["w2o"]).reshape(4, 4), np.eye(4), atol=1e-6)
    d = dump(tmp_path, head + 'Material "none"\nShape "sphere"\nWorldEnd\n')       # a medium boundary
    assert d["spheres"][0]["material"] == abi.NO_MATERIAL
    for body, needle in (('AreaLightSource "spot" "rgb L" [1 1 1]\nShape "sphere"', "not supported (diffuse)"),
                         ('Shape "cylinder"', "not supported")):
        err = dump(tmp_path, head + body + "\nWorldEnd\n", expect_fail=True)
        assert needle in err, err


def test_sinc_pixel_filter_table(tmp_path):
    """PixelFilter "sinc" (CreateSincFilter, Enigma/Filter.cpp:62-68; LanczosSincFilter, Enigma/Filter.h:63-88): radius 4, tau 3 by
    default, the 16 x 16 table of Film's constructor with its negative lobes."""
    text = '''
Camera "perspective"
PixelFilter "sinc" "float tau" 2.5
Film "image" "integer xresolution" 32 "integer yresolution" 16
Sampler "sobol"
Integrator "path"
WorldBegin
Shape "trianglemesh" "point P" [0 0 5  1 0 5  0 1 5]
WorldEnd
'''
    d = dump(tmp_path, text)
    film = scenes.film_desc(32, 16, "sinc", tau=2.5)
    assert d["film"]["filter"] == "sinc"
    table = np.array(d["film"]["filter_table"])
    assert np.allclose(table, list(film.filter_table), rtol=1e-4, atol=1e-6)
    assert table.min() < 0 < table.max()


def test_samplers(tmp_path):
    """MakeSampler's two GlobalSamplers (Enigma/Api.cpp:425-434): halton when the file names none (Enigma/Api.h:82), 16 samples
    by default for both, the Halton sampler's count kept as given and the Sobol' one's rounded up, "samplepixelcenter"."""
    d = dump(tmp_path, 'WorldBegin\nWorldEnd\n')["sampler"]
    assert (d["name"], d["spp"], d["samplepixelcenter"]) == ("halton", 16, False)
    d = dump(tmp_path, 'Sampler "halton" "integer pixelsamples" 5 "bool samplepixelcenter" "true"\nWorldBegin\nWorldEnd\n')["sampler"]
    assert (d["name"], d["spp"], d["samplepixelcenter"]) == ("halton", 5, True)
    d = dump(tmp_path, 'Sampler "sobol" "integer pixelsamples" 5\nWorldBegin\nWorldEnd\n')["sampler"]
    assert (d["name"], d["spp"], d["samplepixelcenter"]) == ("sobol", 8, False)
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nWorldEnd\n')["sampler"]
    assert (d["name"], d["spp"]) == ("sobol", 16)


def test_projection_light(tmp_path):
    """LightSource "projection" (CreateProjectionLight, Enigma/Lights.cpp:782-789): at the CTM's origin, I x scale, the cone of
    "fov" (default 45) at aspect 1; the map is named in a warning because the reference drops it (Enigma/Lights.cpp:218-222)."""
    text = ('Sampler "sobol"\nWorldBegin\nAttributeBegin\nTranslate 1 2 3\nLightSource "projection" "rgb I" [2 4 6] "rgb scale" [0.5 0.5 0.5] "string mapname" "slide.exr"\n'
            'AttributeEnd\nLightSource "projection" "float fov" 90\nLightSource "laser"\nWorldEnd\n')
    err = dump(tmp_path, text, expect_fail=True)
    assert "laser" in err
    d = dump(tmp_path, text.replace('LightSource "laser"\n', ""))
    a, b = d["lights"]
    assert a["type"] == abi.LIGHT_PROJECTION and np.allclose(a["from"], [1, 2, 3]) and np.allclose(a["I"], [1, 2, 3])
    assert np.isclose(a["cos_total_width"], np.cos(np.arctan(np.tan(np.radians(22.5)) * np.sqrt(2.0))), atol=1e-6)
    assert np.allclose(b["from"], [0, 0, 0]) and np.allclose(b["I"], [1, 1, 1]) and np.isclose(b["cos_total_width"], np.cos(np.arctan(np.sqrt(2.0))), atol=1e-6)
    assert any("drops the image" in w for w in d["warnings"])


def test_infinite_light_reads_its_radiance_map_from_a_pfm_file(tmp_path):
    """LightSource "infinite" "string mapname" (CreateInfiniteLight, Enigma/Lights.cpp:801-810): the front end reads PFM files,
    relative to the scene file's directory."""
    import pbrt_writer
    pbrt_writer.write_pfm(str(tmp_path / "sky.pfm"), np.random.RandomState(1).rand(8, 16, 3).astype(np.float32))
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nLightSource "infinite" "rgb L" [2 2 2] "string mapname" "sky.pfm"\nLightSource "infinite"\nWorldEnd\n')
    a, b = d["lights"]
    assert a["type"] == abi.LIGHT_INFINITE and a["map"] == [16, 8] and np.allclose(a["I"], [2, 2, 2])
    assert b["map"] == [0, 0]
    assert "PFM" in dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nLightSource "infinite" "string mapname" "sky.exr"\nWorldEnd\n', expect_fail=True)
    assert "sky2.pfm" in dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nLightSource "infinite" "string mapname" "sky2.pfm"\nWorldEnd\n', expect_fail=True)


def test_goniometric_light(tmp_path):
    """LightSource "goniometric" (CreateGoniometricLight, Enigma/Lights.cpp:774-780): at the CTM's origin, I x scale, its map read from a PFM file."""
    import pbrt_writer
    pbrt_writer.write_pfm(str(tmp_path / "g.pfm"), np.random.RandomState(2).rand(6, 10, 3).astype(np.float32))
    d = dump(tmp_path, 'Sampler "sobol"\nWorldBegin\nAttributeBegin\nTranslate 4 5 6\nLightSource "goniometric" "rgb I" [2 4 6] "rgb scale" [2 2 2] "string mapname" "g.pfm"\n'
                       'AttributeEnd\nLightSource "goniometric"\nWorldEnd\n')
    a, b = d["lights"]
    assert a["type"] == abi.LIGHT_GONIOMETRIC and np.allclose(a["from"], [4, 5, 6]) and np.allclose(a["I"], [4, 8, 12]) and a["map"] == [10, 6]
    assert b["type"] == abi.LIGHT_GONIOMETRIC and np.allclose(b["I"], [1, 1, 1]) and b["map"] == [0, 0]


def test_constant_float_textures_and_scale_and_mix_of_constants(tmp_path):
    """Texture classes that combine others are folded when they are made, with the arithmetic of their Evaluate (ScaleTexture:
    the product, Enigma/Texture.h:147-149; MixTexture: the clamped Lerp, :171-176, Enigma/MathCommon.h:31-39); a float parameter
    bound to a float texture takes its value (GetFloatTexture, Enigma/ParameterSet.cpp:716-743)."""
    import pbrt_writer
    pbrt_writer.write_pfm(str(tmp_path / "img.pfm"), np.ones((2, 2, 3), dtype=np.float32))
    text = """Sampler "sobol"
WorldBegin
Texture "r" "float" "constant" "float value" 0.3
Texture "r2" "float" "scale" "texture tex1" "r" "float tex2" 0.5
Texture "r3" "float" "mix" "texture tex1" "r" "texture tex2" "r2" "float amount" 0.25
Texture "c" "spectrum" "constant" "rgb value" [0.2 0.4 0.8]
Texture "cs" "spectrum" "scale" "texture tex1" "c" "rgb tex2" [0.5 0.5 2]
Texture "cm" "color" "mix" "texture tex1" "c" "texture tex2" "cs" "texture amount" "r3"
Texture "over" "spectrum" "mix" "rgb tex1" [1 1 1] "rgb tex2" [3 3 3] "float amount" 1.5
Material "plastic" "texture Kd" "cs" "texture Ks" "cm" "texture roughness" "r2"
Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]
Material "matte" "texture Kd" "over" "texture sigma" "r3"
Shape "trianglemesh" "point P" [0 0 1 1 0 1 0 1 1]
WorldEnd
"""
    d = dump(tmp_path, text)
    f32 = np.float32
    r, r2 = f32(0.3), f32(0.3) * f32(0.5)
    r3 = r + (r2 - r) * f32(0.25)
    c = np.array([0.2, 0.4, 0.8], dtype=f32)
    cs = c * np.array([0.5, 0.5, 2], dtype=f32)
    cm = c + (cs - c) * r3
    plastic, matte = d["materials"][-2], d["materials"][-1]
    assert np.array_equal(np.array(plastic["kd"], dtype=f32), cs) and np.array_equal(np.array(plastic["ks"], dtype=f32), cm)
    assert f32(plastic["uroughness"]) == r2 and f32(plastic["vroughness"]) == r2
    assert np.array_equal(np.array(matte["kd"], dtype=f32), [3, 3, 3]) and f32(matte["sigma"]) == r3   # an amount above 1 gives the second texture
    assert not [w for w in d["warnings"] if "not used" in w]
    bad = text.replace('Texture "c" "spectrum" "constant" "rgb value" [0.2 0.4 0.8]', 'Texture "c" "spectrum" "imagemap" "string filename" "img.pfm"')
    assert "evaluated per hit" in dump(tmp_path, bad, expect_fail=True)
