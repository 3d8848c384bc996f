"""ctypes mirror of include/enigma_hip.h (POD structs shared by the product library and,
in tests only, by the CPU oracle)."""
import ctypes as C
import numpy as np

MAT_MATTE, MAT_METAL, MAT_GLASS, MAT_PLASTIC = 0, 1, 2, 3
MAT_MIRROR, MAT_TRANSLUCENT, MAT_SUBSTRATE, MAT_UBER, MAT_MIX = 4, 5, 6, 7, 8
LIGHT_POINT, LIGHT_DIFFUSE_AREA, LIGHT_SPOT, LIGHT_DISTANT, LIGHT_INFINITE, LIGHT_DIFFUSE_AREA_SPHERE, LIGHT_PROJECTION, LIGHT_GONIOMETRIC = 0, 1, 2, 3, 4, 5, 6, 7
LIGHTS_UNIFORM, LIGHTS_POWER, LIGHTS_SPATIAL = 0, 1, 2
CAMERA_PERSPECTIVE, CAMERA_ENVIRONMENT = 0, 1
SAMPLER_SOBOL, SAMPLER_HALTON = 0, 1  # enigma_sampler_select
DL_SAMPLE_ALL, DL_SAMPLE_ONE = 0, 1
TRI_FLIP_NORMAL = 1
TRI_HAS_NORMALS = 2
TRI_HAS_UVS = 4
TRI_REVERSE_ORIENTATION = 8
NO_MATERIAL = 0xFFFFFFFF
COMM_ID_BYTES = 128


class EnigmaMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("remap_roughness", C.c_uint32),
                ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("kr", C.c_float * 3), ("kt", C.c_float * 3),
                ("eta", C.c_float * 3), ("k", C.c_float * 3),
                ("sigma", C.c_float), ("uroughness", C.c_float), ("vroughness", C.c_float), ("index", C.c_float),
                ("tex_kd", C.c_uint32), ("tex_ks", C.c_uint32), ("tex_kr", C.c_uint32), ("tex_kt", C.c_uint32),
                ("tex_eta", C.c_uint32), ("tex_k", C.c_uint32),
                ("reflect", C.c_float * 3), ("transmit", C.c_float * 3), ("opacity", C.c_float * 3), ("amount", C.c_float * 3),
                ("mix_a", C.c_uint32), ("mix_b", C.c_uint32)]


WRAP_REPEAT, WRAP_BLACK, WRAP_CLAMP = 0, 1, 2


class EnigmaTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.c_void_p), ("scale", C.c_float), ("gamma", C.c_float),
                ("trilinear", C.c_uint32), ("wrap", C.c_uint32), ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float)]


class EnigmaLight(C.Structure):
    _fields_ = [("type", C.c_uint32), ("triangle", C.c_uint32), ("two_sided", C.c_uint32), ("n_samples", C.c_uint32),
                ("L", C.c_float * 3), ("pos", C.c_float * 3), ("dir", C.c_float * 3),
                ("cos_total_width", C.c_float), ("cos_falloff_start", C.c_float), ("world_to_light", C.c_float * 9),
                ("light_to_world", C.c_float * 9), ("map_width", C.c_uint32), ("map_height", C.c_uint32), ("map_rgb", C.c_void_p)]


class EnigmaMedium(C.Structure):
    _fields_ = [("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3), ("g", C.c_float), ("kind", C.c_uint32)]




class EnigmaSphere(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16),
                ("radius", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("phi_max", C.c_float),
                ("material", C.c_uint32), ("flags", C.c_uint32), ("prim_id", C.c_uint32),
                ("medium_inside", C.c_int32), ("medium_outside", C.c_int32), ("light", C.c_uint32)]


class EnigmaSceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("vertices", C.c_void_p), ("indices", C.c_void_p), ("tri_material", C.c_void_p),
                ("tri_light", C.c_void_p), ("tri_flags", C.c_void_p), ("tri_prim_id", C.c_void_p),
                ("normals", C.c_void_p), ("uvs", C.c_void_p),
                ("n_materials", C.c_uint32), ("n_lights", C.c_uint32),
                ("materials", C.c_void_p), ("lights", C.c_void_p),
                ("n_media", C.c_uint32), ("media", C.c_void_p), ("tri_medium_inside", C.c_void_p), ("tri_medium_outside", C.c_void_p),
                ("n_textures", C.c_uint32), ("textures", C.c_void_p), ("n_spheres", C.c_uint32), ("spheres", C.c_void_p)]


class EnigmaCameraDesc(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float), ("type", C.c_uint32)]


class EnigmaFilmDesc(C.Structure):
    _fields_ = [("full_resolution", C.c_int32 * 2), ("cropped_pixel_bounds", C.c_int32 * 4),
                ("filter_radius", C.c_float * 2), ("filter_table", C.c_float * 256),
                ("max_sample_luminance", C.c_float)]


class EnigmaIntegratorDesc(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("rr_threshold", C.c_float), ("light_strategy", C.c_int32),
                ("pixel_bounds", C.c_int32 * 4), ("samples_per_pixel", C.c_int64), ("sample_bounds", C.c_int32 * 4)]


class EnigmaVolPathDesc(EnigmaIntegratorDesc):
    """Same layout as EnigmaIntegratorDesc; the type selects enigma_volpath_config."""


class EnigmaDirectLightingDesc(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("strategy", C.c_int32), ("pixel_bounds", C.c_int32 * 4),
                ("samples_per_pixel", C.c_int64), ("sample_bounds", C.c_int32 * 4)]


class EnigmaStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_mis", C.c_uint64), ("ms_render", C.c_double), ("ms_intersect", C.c_double),
                ("intersect_launches", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_depth", C.c_uint64),
                ("ms_bvh_build", C.c_double),
                ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64), ("nodes_shadow", C.c_uint64), ("tris_shadow", C.c_uint64),
                ("nodes_mis", C.c_uint64), ("tris_mis", C.c_uint64), ("bvh_node_bytes", C.c_uint64),
                ("scene_in_lds", C.c_uint64), ("ms_shadow", C.c_double), ("shadow_launches", C.c_uint64),
                ("node_steps_closest", C.c_uint64), ("tri_steps_closest", C.c_uint64), ("node_steps_shadow", C.c_uint64), ("tri_steps_shadow", C.c_uint64),
                ("stack_overflows", C.c_uint64), ("ms_shade", C.c_double), ("shade_launches", C.c_uint64), ("shaded_vertices", C.c_uint64)]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class SceneArrays:
    """Owns the numpy arrays an EnigmaSceneDesc points into."""

    def __init__(self, vertices, indices, tri_material, tri_light, materials, lights, tri_flags=None, tri_prim_id=None,
                 media=(), tri_medium_inside=None, tri_medium_outside=None, normals=None, uvs=None, textures=(), spheres=()):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        nt = self.indices.shape[0]
        self.tri_material = np.ascontiguousarray(tri_material, dtype=np.uint32).reshape(nt)
        self.tri_light = np.ascontiguousarray(tri_light, dtype=np.int32).reshape(nt)
        self.tri_flags = None if tri_flags is None else np.ascontiguousarray(tri_flags, dtype=np.uint32).reshape(nt)
        self.tri_prim_id = None if tri_prim_id is None else np.ascontiguousarray(tri_prim_id, dtype=np.uint32).reshape(nt)
        nv = self.vertices.shape[0]
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(nv, 3)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(nv, 2)
        self.media = (EnigmaMedium * max(1, len(media)))(*media)
        self.n_media = len(media)
        self.tri_medium_inside = None if tri_medium_inside is None else np.ascontiguousarray(tri_medium_inside, dtype=np.int32).reshape(nt)
        self.tri_medium_outside = None if tri_medium_outside is None else np.ascontiguousarray(tri_medium_outside, dtype=np.int32).reshape(nt)
        self.materials = (EnigmaMaterial * max(1, len(materials)))(*materials)
        self.textures = (EnigmaTexture * max(1, len(textures)))(*textures)
        self._texture_maps = [getattr(t, "_map", None) for t in textures]   # the arrays rgb points into
        self.n_textures = len(textures)
        self.spheres = (EnigmaSphere * max(1, len(spheres)))(*spheres)
        self.n_spheres = len(spheres)
        self.lights = (EnigmaLight * max(1, len(lights)))(*lights)
        self._light_maps = [getattr(l, "_map", None) for l in lights]   # the arrays map_rgb points into
        self.n_lights = len(lights)
        self.n_materials = len(materials)

    def desc(self):
        d = EnigmaSceneDesc()
        d.n_vertices = self.vertices.shape[0]
        d.n_triangles = self.indices.shape[0]
        d.vertices = ptr(self.vertices)
        d.indices = ptr(self.indices)
        d.tri_material = ptr(self.tri_material)
        d.tri_light = ptr(self.tri_light)
        d.tri_flags = ptr(self.tri_flags)
        d.tri_prim_id = ptr(self.tri_prim_id)
        d.normals = ptr(self.normals)
        d.uvs = ptr(self.uvs)
        d.n_materials = self.n_materials
        d.n_lights = self.n_lights
        d.materials = C.cast(self.materials, C.c_void_p)
        d.lights = C.cast(self.lights, C.c_void_p)
        d.n_media = self.n_media
        d.media = C.cast(self.media, C.c_void_p) if self.n_media else None
        d.tri_medium_inside = ptr(self.tri_medium_inside)
        d.tri_medium_outside = ptr(self.tri_medium_outside)
        d.n_textures = self.n_textures
        d.textures = C.cast(self.textures, C.c_void_p) if self.n_textures else None
        d.n_spheres = self.n_spheres
        d.spheres = C.cast(self.spheres, C.c_void_p) if self.n_spheres else None
        return d


def sphere(material, radius=1.0, object_to_world=None, z_min=None, z_max=None, phi_max=360.0, reverse_orientation=False,
           medium_inside=-1, medium_outside=-1):
    """CreateSphereShape (Enigma/ShapeImp.cpp) under a given object-to-world matrix (4 x 4, row-major; identity if omitted).
    The inverse is taken in double and rounded once, as the scene-description front end does."""
    s = EnigmaSphere()
    m = np.eye(4) if object_to_world is None else np.asarray(object_to_world, dtype=np.float64).reshape(4, 4)
    s.object_to_world[:] = m.astype(np.float32).reshape(-1).tolist()
    s.world_to_object[:] = np.linalg.inv(m).astype(np.float32).reshape(-1).tolist()
    s.radius = radius
    s.z_min = -radius if z_min is None else z_min
    s.z_max = radius if z_max is None else z_max
    s.phi_max = phi_max
    s.material = material
    s.medium_inside, s.medium_outside = medium_inside + 1, medium_outside + 1   # the ABI stores index + 1, 0 = none
    swaps = np.linalg.det(m[:3, :3]) < 0
    s.flags = (TRI_FLIP_NORMAL if (bool(reverse_orientation) != bool(swaps)) else 0) | (TRI_REVERSE_ORIENTATION if reverse_orientation else 0)
    return s


def sphere_area_light(sphere_index, L, two_sided=False, n_samples=1):
    """The DiffuseAreaLight of an emissive sphere; the sphere's `light` field must hold this light's index + 1."""
    l = area_light(sphere_index, L, two_sided, n_samples)
    l.type = LIGHT_DIFFUSE_AREA_SPHERE
    return l


def image_texture(rgb, trilinear=False, gamma=True, scale=1.0, wrap=WRAP_REPEAT, su=1.0, sv=1.0, du=0.0, dv=0.0):
    """CreateImageSpectrumTexture (Enigma/Texture.cpp:412-453) for texels already read: rgb is [height, width, 3],
    top row first.  gamma is the reference's bool (it ends up as the exponent 1 or 0 of the conversion)."""
    t = EnigmaTexture()
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    assert a.ndim == 3 and a.shape[2] == 3
    t._map = a
    t.height, t.width = a.shape[0], a.shape[1]
    t.rgb = a.ctypes.data_as(C.c_void_p)
    t.scale, t.gamma = scale, 1.0 if gamma else 0.0
    t.trilinear, t.wrap = int(bool(trilinear)), wrap
    t.su, t.sv, t.du, t.dv = su, sv, du, dv
    return t


def matte(kd, sigma=0.0):
    m = EnigmaMaterial()
    m.type = MAT_MATTE
    m.kd[:] = kd
    m.sigma = sigma
    return m


def metal(eta, k, roughness=0.01, remap=True, uroughness=None, vroughness=None):
    m = EnigmaMaterial()
    m.type = MAT_METAL
    m.eta[:] = eta
    m.k[:] = k
    m.uroughness = roughness if uroughness is None else uroughness
    m.vroughness = roughness if vroughness is None else vroughness
    m.remap_roughness = 1 if remap else 0
    return m


def glass(index=1.5, uroughness=0.0, vroughness=0.0, kr=(1, 1, 1), kt=(1, 1, 1), remap=True):
    m = EnigmaMaterial()
    m.type = MAT_GLASS
    m.kr[:] = kr
    m.kt[:] = kt
    m.index = index
    m.uroughness = uroughness
    m.vroughness = vroughness
    m.remap_roughness = 1 if remap else 0
    return m


def plastic(kd, ks, roughness=0.1, remap=True):
    m = EnigmaMaterial()
    m.type = MAT_PLASTIC
    m.kd[:] = kd
    m.ks[:] = ks
    m.uroughness = roughness
    m.vroughness = roughness
    m.remap_roughness = 1 if remap else 0
    return m


def mirror(kr=(0.9, 0.9, 0.9)):
    m = EnigmaMaterial()
    m.type = MAT_MIRROR
    m.kr[:] = kr
    return m


def translucent(kd=(0.25,) * 3, ks=(0.25,) * 3, reflect=(0.5,) * 3, transmit=(0.5,) * 3, roughness=0.1, remap=True):
    m = EnigmaMaterial()
    m.type = MAT_TRANSLUCENT
    m.kd[:] = kd
    m.ks[:] = ks
    m.reflect[:] = reflect
    m.transmit[:] = transmit
    m.uroughness = m.vroughness = roughness
    m.remap_roughness = 1 if remap else 0
    return m


def substrate(kd=(0.5,) * 3, ks=(0.5,) * 3, uroughness=0.1, vroughness=0.1, remap=True):
    m = EnigmaMaterial()
    m.type = MAT_SUBSTRATE
    m.kd[:] = kd
    m.ks[:] = ks
    m.uroughness, m.vroughness = uroughness, vroughness
    m.remap_roughness = 1 if remap else 0
    return m


def uber(kd=(0.25,) * 3, ks=(0.25,) * 3, kr=(0, 0, 0), kt=(0, 0, 0), opacity=(1, 1, 1), roughness=0.1, uroughness=None, vroughness=None,
         index=1.5, remap=True):
    """uroughness / vroughness as UberMaterial resolves them (Enigma/Material.cpp:563-570): u = uroughness or roughness, v = vroughness or u."""
    m = EnigmaMaterial()
    m.type = MAT_UBER
    m.kd[:] = kd
    m.ks[:] = ks
    m.kr[:] = kr
    m.kt[:] = kt
    m.opacity[:] = opacity
    m.uroughness = roughness if uroughness is None else uroughness
    m.vroughness = m.uroughness if vroughness is None else vroughness
    m.index = index
    m.remap_roughness = 1 if remap else 0
    return m


def mix(a, b, amount=(0.5, 0.5, 0.5)):
    """a, b: indices of the two materials in the scene's material array, both before the mix itself."""
    m = EnigmaMaterial()
    m.type = MAT_MIX
    m.mix_a, m.mix_b = a, b
    m.amount[:] = amount
    return m


MEDIUM_HOMOGENEOUS, MEDIUM_GRID_DENSITY = 0, 1


def homogeneous_medium(sigma_a, sigma_s, g=0.0):
    m = EnigmaMedium()
    m.sigma_a[:] = sigma_a
    m.sigma_s[:] = sigma_s
    m.g = g
    return m


def grid_density_medium(sigma_a, sigma_s, g=0.0):
    """The reference's GridDensityMedium as it stands: Tr and Sample return black (Enigma/Medium.cpp:51-59)."""
    m = homogeneous_medium(sigma_a, sigma_s, g)
    m.kind = MEDIUM_GRID_DENSITY
    return m


def area_light(triangle, L, two_sided=False, n_samples=1):
    l = EnigmaLight()
    l.type = LIGHT_DIFFUSE_AREA
    l.triangle = triangle
    l.two_sided = 1 if two_sided else 0
    l.n_samples = n_samples
    l.L[:] = L
    return l


def point_light(pos, I, n_samples=1):
    l = EnigmaLight()
    l.type = LIGHT_POINT
    l.n_samples = n_samples
    l.pos[:] = pos
    l.L[:] = I
    return l


def projection_light(pos, I, fov=45.0, n_samples=1):
    """ProjectionLight as the reference has it (Enigma/Lights.cpp:194-296): no map survives its constructor, so it shines like a
    point light; the cone of its field of view only sizes Power()."""
    f32 = np.float32
    l = point_light(pos, I, n_samples)
    l.type = LIGHT_PROJECTION
    opposite = f32(np.tan(f32(f32(f32(fov) * f32(np.pi / 180.0)) / f32(2))))
    tan_diag = f32(opposite * f32(np.sqrt(f32(2))))
    l.cos_total_width = float(f32(np.cos(f32(np.arctan(tan_diag)))))
    return l


def coordinate_system(v1):
    """CoordinateSystem (Enigma/MathCommon.h:371-382) in float32."""
    f = np.float32
    v1 = np.asarray(v1, dtype=f)
    if abs(v1[0]) > abs(v1[1]):
        inv = f(1) / np.sqrt(v1[0] * v1[0] + v1[2] * v1[2])
        v2 = np.array([-v1[2] * inv, 0, v1[0] * inv], dtype=f)
    else:
        inv = f(1) / np.sqrt(v1[1] * v1[1] + v1[2] * v1[2])
        v2 = np.array([0, v1[2] * inv, -v1[1] * inv], dtype=f)
    v3 = np.array([v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]], dtype=f)
    return v2, v3


def spot_light(pos, to, I, coneangle=30.0, conedeltaangle=5.0, n_samples=1):
    """CreateSpotLight (Enigma/Lights.cpp:752-771) under an identity CTM: the light looks from pos towards
    `to`; world-to-light is the rotation whose rows are (du, dv, dir)."""
    f = np.float32
    d = np.asarray(to, dtype=f) - np.asarray(pos, dtype=f)
    d = d / np.sqrt((d * d).sum(dtype=f))
    du, dv = coordinate_system(d)
    l = EnigmaLight()
    l.type = LIGHT_SPOT
    l.n_samples = n_samples
    l.pos[:] = pos
    l.L[:] = I
    l.cos_total_width = float(np.cos(f(np.radians(f(coneangle)))))
    l.cos_falloff_start = float(np.cos(f(np.radians(f(coneangle - conedeltaangle)))))
    l.world_to_light[:] = [float(x) for x in np.concatenate([du, dv, d])]
    return l


def infinite_light(L, radiance_map=None, light_to_world=None, n_samples=1):
    """InfiniteAreaLight (Enigma/Lights.cpp:395-442): radiance_map is a (height, width, 3) float32 latitude-longitude
    image (power-of-two sizes) or None for the constant environment L; light_to_world a 3x3 rotation (default identity)."""
    l = EnigmaLight()
    l.type = LIGHT_INFINITE
    l.n_samples = n_samples
    l.L[:] = L
    m = np.eye(3, dtype=np.float64) if light_to_world is None else np.asarray(light_to_world, dtype=np.float64).reshape(3, 3)
    l.light_to_world[:] = [float(np.float32(x)) for x in m.reshape(-1)]
    l.world_to_light[:] = [float(np.float32(x)) for x in np.linalg.inv(m).reshape(-1)]
    if radiance_map is not None:
        a = np.ascontiguousarray(radiance_map, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 3
        l.map_height, l.map_width = a.shape[0], a.shape[1]
        l.map_rgb = a.ctypes.data_as(C.c_void_p)
        l._map = a
    return l


def goniometric_light(pos, I, gonio_map=None, n_samples=1):
    """GonioPhotometricLight as the reference has it (Enigma/Lights.cpp:301-363): a point light of intensity I x the texel at the
    top of the map's image pyramid; gonio_map is a (height, width, 3) float32 image or None."""
    l = point_light(pos, I, n_samples)
    l.type = LIGHT_GONIOMETRIC
    if gonio_map is not None:
        a = np.ascontiguousarray(gonio_map, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 3
        l.map_height, l.map_width = a.shape[0], a.shape[1]
        l.map_rgb = a.ctypes.data_as(C.c_void_p)
        l._map = a
    return l


def distant_light(frm, to, L, n_samples=1):
    """CreateDistantLight (Enigma/Lights.cpp:791-799) under an identity CTM: m_lightDir = from - to, kept
    at whatever length it has."""
    l = EnigmaLight()
    l.type = LIGHT_DISTANT
    l.n_samples = n_samples
    l.dir[:] = [float(np.float32(a) - np.float32(b)) for a, b in zip(frm, to)]
    l.L[:] = L
    return l
