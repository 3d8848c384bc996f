"""Writes a pbrt-format scene description for a SceneArrays scene (test helper for the scene-description
front end).  One trianglemesh Shape per run of triangles that share material and emission; materials as
MakeNamedMaterial.  Light order in the file = light order in SceneArrays (lights are created in file order)."""
import ctypes as C
import math
import os

import numpy as np

from raytrace_amd import abi


def _rgb(name, v):
    return '"rgb %s" [%.9g %.9g %.9g]' % (name, v[0], v[1], v[2])


def write_pfm(path, rgb):
    """rgb: [height, width, 3], top row first.  PFM stores rows bottom to top; a negative scale marks little endian."""
    a = np.ascontiguousarray(rgb, dtype="<f4")
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0]))
        f.write(a[::-1].tobytes())


def texture_text(i, t, directory):
    """One Texture directive per image texture; the texels go into <directory>/tex<i>.pfm."""
    rgb = np.ctypeslib.as_array(C.cast(t.rgb, C.POINTER(C.c_float)), shape=(t.height, t.width, 3))
    write_pfm(os.path.join(directory, "tex%d.pfm" % i), rgb)
    wrap = {abi.WRAP_REPEAT: "repeat", abi.WRAP_BLACK: "black", abi.WRAP_CLAMP: "clamp"}[t.wrap]
    return ('Texture "t%d" "spectrum" "imagemap" "string filename" "tex%d.pfm" "bool trilinear" "%s" "bool gamma" "%s" "float scale" %.9g '
            '"string wrap" "%s" "float uscale" %.9g "float vscale" %.9g "float udelta" %.9g "float vdelta" %.9g' %
            (i, i, "true" if t.trilinear else "false", "true" if t.gamma != 0 else "false", t.scale, wrap, t.su, t.sv, t.du, t.dv))


def write_ply(path, verts, tris, normals=None, uvs=None, fmt="binary_little_endian", quads=None):
    """A PLY file of a triangle mesh (quads: optional [n, 4] faces written after the triangles)."""
    import struct
    n = len(verts)
    head = ["ply", "format %s 1.0" % fmt, "comment written by tests/pbrt_writer.py", "element vertex %d" % n,
            "property float x", "property float y", "property float z"]
    if normals is not None:
        head += ["property float nx", "property float ny", "property float nz"]
    if uvs is not None:
        head += ["property float u", "property float v"]
    nq = 0 if quads is None else len(quads)
    head += ["element face %d" % (len(tris) + nq), "property list uchar int vertex_indices", "end_header"]
    cols = [np.asarray(verts, dtype=np.float32)]
    if normals is not None:
        cols.append(np.asarray(normals, dtype=np.float32))
    if uvs is not None:
        cols.append(np.asarray(uvs, dtype=np.float32))
    table = np.concatenate(cols, axis=1)
    faces = [list(map(int, t)) for t in tris] + ([] if quads is None else [list(map(int, q)) for q in quads])
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if fmt == "ascii":
            for row in table:
                f.write((" ".join("%.9g" % float(v) for v in row) + "\n").encode())
            for fc in faces:
                f.write(("%d %s\n" % (len(fc), " ".join(str(v) for v in fc))).encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            f.write(table.astype(e + "f4").tobytes())
            for fc in faces:
                f.write(struct.pack(e + "B%di" % len(fc), len(fc), *fc))


def material_text(i, m):
    t = {abi.MAT_MATTE: "matte", abi.MAT_METAL: "metal", abi.MAT_GLASS: "glass", abi.MAT_PLASTIC: "plastic", abi.MAT_MIRROR: "mirror",
         abi.MAT_TRANSLUCENT: "translucent", abi.MAT_SUBSTRATE: "substrate", abi.MAT_UBER: "uber", abi.MAT_MIX: "mix"}[m.type]
    parts = ['MakeNamedMaterial "m%d" "string type" "%s"' % (i, t)]
    for name, tex in (("Kd", m.tex_kd), ("Ks", m.tex_ks), ("Kr", m.tex_kr), ("Kt", m.tex_kt), ("eta", m.tex_eta), ("k", m.tex_k)):
        if tex:
            parts.append('"texture %s" "t%d"' % (name, tex - 1))
    remap = '"bool remaproughness" "%s"' % ("true" if m.remap_roughness else "false")
    pre = ""
    if t == "matte" and i % 2 == 1:   # every other one through a float texture: a scale of two constants whose product is the value
        pre = ('Texture "fs%d" "float" "constant" "float value" %.9g\nTexture "fsig%d" "float" "scale" "texture tex1" "fs%d" "float tex2" 1\n' % (i, m.sigma, i, i))
        parts += [_rgb("Kd", m.kd), '"texture sigma" "fsig%d"' % i]
    elif t == "matte":
        parts += [_rgb("Kd", m.kd), '"float sigma" %.9g' % m.sigma]
    elif t == "metal":
        parts += [_rgb("eta", m.eta), _rgb("k", m.k), '"float uroughness" %.9g' % m.uroughness, '"float vroughness" %.9g' % m.vroughness, remap]
    elif t == "glass":
        parts += [_rgb("Kr", m.kr), _rgb("Kt", m.kt), '"float index" %.9g' % m.index, '"float uroughness" %.9g' % m.uroughness,
                  '"float vroughness" %.9g' % m.vroughness, remap]
    elif t == "plastic" and i % 2 == 1:
        pre = 'Texture "fr%d" "float" "constant" "float value" %.9g\n' % (i, m.uroughness)
        parts += [_rgb("Kd", m.kd), _rgb("Ks", m.ks), '"texture roughness" "fr%d"' % i, remap]
    elif t == "plastic":
        parts += [_rgb("Kd", m.kd), _rgb("Ks", m.ks), '"float roughness" %.9g' % m.uroughness, remap]
    elif t == "mirror":
        parts += [_rgb("Kr", m.kr)]
    elif t == "translucent":
        parts += [_rgb("Kd", m.kd), _rgb("Ks", m.ks), _rgb("reflect", m.reflect), _rgb("transmit", m.transmit), '"float roughness" %.9g' % m.uroughness, remap]
    elif t == "substrate":
        parts += [_rgb("Kd", m.kd), _rgb("Ks", m.ks), '"float uroughness" %.9g' % m.uroughness, '"float vroughness" %.9g' % m.vroughness, remap]
    elif t == "uber":
        parts += [_rgb("Kd", m.kd), _rgb("Ks", m.ks), _rgb("Kr", m.kr), _rgb("Kt", m.kt), _rgb("opacity", m.opacity), '"float uroughness" %.9g' % m.uroughness,
                  '"float vroughness" %.9g' % m.vroughness, '"float index" %.9g' % m.index, remap]
    else:
        parts += ['"string namedmaterial1" "m%d"' % m.mix_a, '"string namedmaterial2" "m%d"' % m.mix_b, _rgb("amount", m.amount)]
    return pre + " ".join(parts)


def scene_text(sc, eye, look, up, fov, xres, yres, spp, integrator='Integrator "path" "integer maxdepth" 5 "string lightsamplestrategy" "power"',
               pixel_filter='PixelFilter "box"', film_extra="", world_prefix="", lens="", texture_dir=None, ply_dir=None, camera="perspective",
               sampler='Sampler "sobol"'):
    """sampler: the directive without its sample count; None leaves the line out (the reference then makes a halton sampler of 16 samples).
    ply_dir: the meshes go into <ply_dir>/mesh<k>.ply (formats in turn) and the scene names them with Shape "plymesh"."""
    out = ["# written by tests/pbrt_writer.py",
           "LookAt %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g" % (tuple(eye) + tuple(look) + tuple(up)),
           'Camera "perspective" "float fov" %.9g %s' % (fov, lens) if camera == "perspective" else 'Camera "%s"' % camera,
           'Film "image" "integer xresolution" [%d] "integer yresolution" [%d] %s' % (xres, yres, film_extra),
           ('%s "integer pixelsamples" %d' % (sampler, spp)) if sampler else "# no Sampler directive", pixel_filter, integrator, "WorldBegin"]
    if world_prefix:
        out.append(world_prefix)
    for i in range(getattr(sc, "n_textures", 0)):
        out.append(texture_text(i, sc.textures[i], texture_dir))
    for i in range(sc.n_materials):
        out.append(material_text(i, sc.materials[i]))
    n_media = getattr(sc, "n_media", 0)
    for i in range(n_media):
        m = sc.media[i]
        out.append('MakeNamedMedium "med%d" "string type" "homogeneous" %s %s "float g" %.9g' % (i, _rgb("sigma_a", m.sigma_a), _rgb("sigma_s", m.sigma_s), m.g))
    med = lambda k: (int(sc.tri_medium_inside[k]), int(sc.tri_medium_outside[k])) if n_media else (-1, -1)
    nT = sc.indices.shape[0]
    # point lights keep their position among the area lights: emit each before the first triangle whose light index is larger
    point = [(i, sc.lights[i]) for i in range(sc.n_lights) if sc.lights[i].type not in (abi.LIGHT_DIFFUSE_AREA, abi.LIGHT_DIFFUSE_AREA_SPHERE)]
    # spheres are written last, so their lights must be the last ones and in sphere order
    on_spheres = [i for i in range(sc.n_lights) if sc.lights[i].type == abi.LIGHT_DIFFUSE_AREA_SPHERE]
    assert on_spheres == list(range(sc.n_lights - len(on_spheres), sc.n_lights)) and [sc.lights[i].triangle for i in on_spheres] == sorted(sc.lights[i].triangle for i in on_spheres)
    emitted = 0

    def flush_points(upto):
        nonlocal emitted
        while emitted < len(point) and point[emitted][0] < upto:
            l = point[emitted][1]
            if l.type == abi.LIGHT_POINT:
                out.append('LightSource "point" %s "point from" [%.9g %.9g %.9g]' % (_rgb("I", l.L), l.pos[0], l.pos[1], l.pos[2]))
            elif l.type == abi.LIGHT_PROJECTION:   # placed by the CTM; the field of view back from the cosine of its cone
                fov = 2 * math.degrees(math.atan(math.tan(math.acos(l.cos_total_width)) / math.sqrt(2.0)))
                out.append('AttributeBegin\n  Translate %.9g %.9g %.9g\n  LightSource "projection" %s "float fov" %.9g "string mapname" "slide.pfm"\nAttributeEnd' %
                           (l.pos[0], l.pos[1], l.pos[2], _rgb("I", l.L), fov))
            elif l.type == abi.LIGHT_GONIOMETRIC:  # placed by the CTM, its map in <texture_dir>/gonio<k>.pfm
                mapname = ""
                if l.map_rgb:
                    texels = np.ctypeslib.as_array(C.cast(l.map_rgb, C.POINTER(C.c_float)), shape=(l.map_height, l.map_width, 3))
                    write_pfm(os.path.join(texture_dir, "gonio%d.pfm" % emitted), texels)
                    mapname = ' "string mapname" "gonio%d.pfm"' % emitted
                out.append('AttributeBegin\n  Translate %.9g %.9g %.9g\n  LightSource "goniometric" %s%s\nAttributeEnd' % (l.pos[0], l.pos[1], l.pos[2], _rgb("I", l.L), mapname))
            elif l.type == abi.LIGHT_SPOT:   # the light looks along the third row of its world-to-light rotation
                d = [l.world_to_light[6], l.world_to_light[7], l.world_to_light[8]]
                cone = math.degrees(math.acos(l.cos_total_width))
                start = math.degrees(math.acos(l.cos_falloff_start))
                out.append('LightSource "spot" %s "point from" [%.9g %.9g %.9g] "point to" [%.9g %.9g %.9g] "float coneangle" %.9g "float conedeltaangle" %.9g' %
                           (_rgb("I", l.L), l.pos[0], l.pos[1], l.pos[2], l.pos[0] + 100 * d[0], l.pos[1] + 100 * d[1], l.pos[2] + 100 * d[2], cone, cone - start))
            elif l.type == abi.LIGHT_INFINITE:      # the rotation goes into the CTM, a radiance map into <texture_dir>/env<k>.pfm
                m = [l.light_to_world[k] for k in range(9)]
                mapname = ""
                if l.map_rgb:
                    texels = np.ctypeslib.as_array(C.cast(l.map_rgb, C.POINTER(C.c_float)), shape=(l.map_height, l.map_width, 3))
                    write_pfm(os.path.join(texture_dir, "env%d.pfm" % emitted), texels)
                    mapname = ' "string mapname" "env%d.pfm"' % emitted
                out.append('AttributeBegin\n  ConcatTransform [%.9g %.9g %.9g 0  %.9g %.9g %.9g 0  %.9g %.9g %.9g 0  0 0 0 1]\n  LightSource "infinite" %s%s\nAttributeEnd' %
                           (m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8], _rgb("L", l.L), mapname))
            elif l.type == abi.LIGHT_DISTANT:
                out.append('LightSource "distant" %s "point from" [%.9g %.9g %.9g] "point to" [0 0 0]' % (_rgb("L", l.L), l.dir[0], l.dir[1], l.dir[2]))
            emitted += 1
    t = 0
    while t < nT:
        e = t
        li = int(sc.tri_light[t])
        key = lambda k: (int(sc.tri_material[k]), int(sc.tri_light[k]) >= 0,
                         tuple(sc.lights[sc.tri_light[k]].L) + (sc.lights[sc.tri_light[k]].two_sided, sc.lights[sc.tri_light[k]].n_samples) if sc.tri_light[k] >= 0 else None,
                         int(sc.tri_flags[k]) if sc.tri_flags is not None else 0, med(k))
        while e < nT and key(e) == key(t) and (li < 0 or int(sc.tri_light[e]) == li + (e - t)):
            e += 1
        if li >= 0:
            flush_points(li)
        verts = sc.vertices[sc.indices[t:e].reshape(-1)]
        idx = np.arange(verts.shape[0])
        out.append("AttributeBegin")
        if int(sc.tri_material[t]) == abi.NO_MATERIAL:
            out.append('  Material "none"')
        else:
            out.append('  NamedMaterial "m%d"' % int(sc.tri_material[t]))
        if med(t) != (-1, -1):
            out.append('  MediumInterface "%s" "%s"' % tuple("" if i < 0 else "med%d" % i for i in med(t)))
        if sc.tri_flags is not None and int(sc.tri_flags[t]) & 1:
            out.append("  ReverseOrientation")
        if li >= 0:
            l = sc.lights[li]
            out.append('  AreaLightSource "diffuse" %s "bool twosided" "%s" "integer samples" %d' % (_rgb("L", l.L), "true" if l.two_sided else "false", max(1, l.n_samples)))
        fl = int(sc.tri_flags[t]) if sc.tri_flags is not None else 0
        extra = ""
        if fl & abi.TRI_HAS_NORMALS:
            extra += ' "normal N" [%s]' % " ".join("%.9g" % float(v) for v in sc.normals[sc.indices[t:e].reshape(-1)].reshape(-1))
        if fl & abi.TRI_HAS_UVS:
            extra += ' "float uv" [%s]' % " ".join("%.9g" % float(v) for v in sc.uvs[sc.indices[t:e].reshape(-1)].reshape(-1))
        if ply_dir is not None:
            k = sum(1 for line in out if 'Shape "plymesh"' in line)
            ids = sc.indices[t:e].reshape(-1)
            write_ply(os.path.join(ply_dir, "mesh%d.ply" % k), verts, idx.reshape(-1, 3), normals=sc.normals[ids] if fl & abi.TRI_HAS_NORMALS else None,
                      uvs=sc.uvs[ids] if fl & abi.TRI_HAS_UVS else None, fmt=("binary_little_endian", "ascii", "binary_big_endian")[k % 3])
            out.append('  Shape "plymesh" "string filename" "mesh%d.ply"' % k)
            out.append("AttributeEnd")
            t = e
            continue
        out.append('  Shape "trianglemesh" "integer indices" [%s] "point P" [%s]%s' %
                   (" ".join(str(int(v)) for v in idx), " ".join("%.9g" % float(v) for v in verts.reshape(-1)), extra))
        out.append("AttributeEnd")
        t = e
    flush_points(1 << 30)
    for i in range(getattr(sc, "n_spheres", 0)):
        s = sc.spheres[i]
        m = np.array(list(s.object_to_world), dtype=np.float64).reshape(4, 4)
        swaps = np.linalg.det(m[:3, :3]) < 0
        reverse = bool(s.flags & abi.TRI_FLIP_NORMAL) != bool(swaps)
        out.append("AttributeBegin")
        out.append("  ConcatTransform [%s]" % " ".join("%.9g" % v for v in m.T.reshape(-1)))      # the file holds the matrix column by column
        out.append('  Material "none"' if s.material == abi.NO_MATERIAL else '  NamedMaterial "m%d"' % s.material)
        if n_media and (s.medium_inside or s.medium_outside):
            out.append('  MediumInterface "%s" "%s"' % tuple("" if i <= 0 else "med%d" % (i - 1) for i in (s.medium_inside, s.medium_outside)))
        if reverse:
            out.append("  ReverseOrientation")
        if s.light:
            l = sc.lights[s.light - 1]
            out.append('  AreaLightSource "diffuse" %s "bool twosided" "%s" "integer samples" %d' % (_rgb("L", l.L), "true" if l.two_sided else "false", max(1, l.n_samples)))
        out.append('  Shape "sphere" "float radius" %.9g "float zmin" %.9g "float zmax" %.9g "float phimax" %.9g' % (s.radius, s.z_min, s.z_max, s.phi_max))
        out.append("AttributeEnd")
    out.append("WorldEnd")
    return "\n".join(out) + "\n"
