"""Deterministic synthetic inputs for the PathIntegrator path.

The reference ships no scene the ray tracer can load (its only call site asks for
scenes/cornell-box/scene.pbrt, Enigma/Main.h:240-246, which is absent), so the fixtures are
authored here: the classic Cornell-box data as triangle meshes, a procedural >= 1 M triangle
scene and a many-light variant.  Everything is world space with final numeric matrices, which
is what the C ABI takes (transform composition of the reference is not reproduced).
"""
import math
import numpy as np

from . import abi

F32 = np.float32


# ----------------------------------------------------------------------------- camera / film
def look_at(pos, look, up):
    """camera-to-world, left-handed, +z forward (the numeric result of Inverse(LookAt))."""
    pos, look, up = (np.asarray(v, dtype=np.float64) for v in (pos, look, up))
    d = look - pos
    d /= np.linalg.norm(d)
    right = np.cross(up / np.linalg.norm(up), d)
    right /= np.linalg.norm(right)
    new_up = np.cross(d, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, new_up, d, pos
    return m


def perspective_raster_to_camera(fov_deg, xres, yres, n=1e-2, f=1000.0):
    """rasterToCamera = Inverse(cameraToScreen) * Inverse(screenToRaster)
    (Enigma/Camera.cpp:90-97,421-430) evaluated in float64."""
    # the screen window is a single-precision quantity (CreatePerspectiveCamera, Enigma/Camera.cpp:385-432: float frame,
    # Bounds2f screen); everything after it is evaluated in float64 and rounded once
    frame = F32(xres) / F32(yres)
    if frame > F32(1.0):
        screen = (float(-frame), float(frame), -1.0, 1.0)
    else:
        screen = (-1.0, 1.0, float(F32(-1.0) / frame), float(F32(1.0) / frame))
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], dtype=np.float64)
    inv_tan = 1.0 / math.tan(math.radians(fov_deg) / 2)
    cam_to_screen = np.diag([inv_tan, inv_tan, 1, 1]) @ persp
    tr = np.eye(4)
    tr[0, 3], tr[1, 3] = -screen[0], -screen[3]
    screen_to_raster = np.diag([xres, yres, 1, 1]) @ np.diag([1 / (screen[1] - screen[0]), 1 / (screen[2] - screen[3]), 1, 1]) @ tr
    return np.linalg.inv(cam_to_screen) @ np.linalg.inv(screen_to_raster)


def environment_camera_desc(pos, look, up):
    """EnvironmentCamera: every direction around pos; the camera's z axis (towards look) is the top row of the film."""
    c = abi.EnigmaCameraDesc()
    c.raster_to_camera[:] = np.eye(4, dtype=F32).reshape(-1).tolist()
    c.camera_to_world[:] = look_at(pos, look, up).astype(F32).reshape(-1).tolist()
    c.focal_distance = 1e30
    c.type = abi.CAMERA_ENVIRONMENT
    return c


def camera_desc(pos, look, up, fov_deg, xres, yres, lens_radius=0.0, focal_distance=1e30):
    c = abi.EnigmaCameraDesc()
    r2c = perspective_raster_to_camera(fov_deg, xres, yres).astype(F32)
    c2w = look_at(pos, look, up).astype(F32)
    c.raster_to_camera[:] = r2c.reshape(-1).tolist()
    c.camera_to_world[:] = c2w.reshape(-1).tolist()
    c.lens_radius = lens_radius
    c.focal_distance = focal_distance
    return c


def _filter_eval(kind, x, y, rx, ry, params):
    """Filter::Evaluate (Enigma/Filter.cpp:8-32, Enigma/Filter.h:50-103) in binary32."""
    x, y, rx, ry = F32(x), F32(y), F32(rx), F32(ry)
    if kind == "box":
        return F32(1.0)
    if kind == "triangle":
        return F32(max(F32(0), rx - abs(x)) * max(F32(0), ry - abs(y)))
    if kind == "gaussian":
        alpha = F32(params.get("alpha", 2.0))
        ex, ey = F32(np.exp(-alpha * rx * rx)), F32(np.exp(-alpha * ry * ry))
        g = lambda d, e: max(F32(0), F32(F32(np.exp(-alpha * d * d)) - e))
        return F32(g(x, ex) * g(y, ey))
    if kind == "mitchell":
        b, c = F32(params.get("B", 1.0 / 3.0)), F32(params.get("C", 1.0 / 3.0))

        def m1(v):
            v = F32(abs(F32(2) * v))
            if v > 1:
                return F32(((-b - 6 * c) * v * v * v + (6 * b + 30 * c) * v * v + (-12 * b - 48 * c) * v + (8 * b + 24 * c)) * F32(1.0 / 6.0))
            return F32(((12 - 9 * b - 6 * c) * v * v * v + (-18 + 12 * b + 6 * c) * v * v + (6 - 2 * b)) * F32(1.0 / 6.0))
        return F32(m1(x * (F32(1) / rx)) * m1(y * (F32(1) / ry)))
    if kind == "sinc":      # LanczosSincFilter::Evaluate / Sinc1D (Enigma/Filter.cpp:18-21, Enigma/Filter.h:74-82)
        tau = F32(params.get("tau", 3.0))

        def s1(v):
            v = F32(abs(v))
            if v < 1e-5:
                return F32(1)
            if v > 1.0:
                return F32(0)
            v = F32(v * F32(math.pi))
            sinc = F32(F32(np.sin(v)) / v)
            lanczos = F32(F32(np.sin(F32(v * tau))) / F32(v * tau))
            return F32(sinc * lanczos)
        return F32(s1(x * (F32(1) / rx)) * s1(y * (F32(1) / ry)))
    raise ValueError(kind)


def film_desc(xres, yres, filter_kind="box", radius=None, crop=(0.0, 1.0, 0.0, 1.0), max_sample_luminance=float("inf"), **params):
    """Film constructor state (Enigma/Film.cpp:39-71): cropped pixel bounds + 16x16 filter table."""
    default_r = {"box": 0.5, "triangle": 2.0, "gaussian": 2.0, "mitchell": 2.0, "sinc": 4.0}[filter_kind]
    rx, ry = (default_r, default_r) if radius is None else radius
    f = abi.EnigmaFilmDesc()
    f.full_resolution[:] = [xres, yres]
    f.cropped_pixel_bounds[:] = [int(math.ceil(xres * crop[0])), int(math.ceil(yres * crop[2])),
                                 int(math.ceil(xres * crop[1])), int(math.ceil(yres * crop[3]))]
    f.filter_radius[:] = [rx, ry]
    tab = []
    for y in range(16):
        for x in range(16):
            px = F32(F32(x + 0.5) * F32(rx) / F32(16))
            py = F32(F32(y + 0.5) * F32(ry) / F32(16))
            tab.append(float(_filter_eval(filter_kind, px, py, rx, ry, params)))
    f.filter_table[:] = tab
    f.max_sample_luminance = max_sample_luminance
    return f


def sample_bounds(film):
    """Film::GetSampleBounds (Enigma/Film.cpp:73-81)."""
    b = film.cropped_pixel_bounds
    rx, ry = F32(film.filter_radius[0]), F32(film.filter_radius[1])
    return [int(math.floor(F32(F32(b[0]) + F32(0.5) - rx))), int(math.floor(F32(F32(b[1]) + F32(0.5) - ry))),
            int(math.ceil(F32(F32(b[2]) - F32(0.5) + rx))), int(math.ceil(F32(F32(b[3]) - F32(0.5) + ry)))]


def integrator_desc(film, spp, max_depth=5, rr_threshold=1.0, light_strategy=abi.LIGHTS_POWER, pixel_bounds=None):
    d = abi.EnigmaIntegratorDesc()
    sb = sample_bounds(film)
    d.max_depth = max_depth
    d.rr_threshold = rr_threshold
    d.light_strategy = light_strategy
    d.pixel_bounds[:] = sb if pixel_bounds is None else pixel_bounds
    d.samples_per_pixel = spp
    d.sample_bounds[:] = sb
    return d


def volpath_desc(film, spp, max_depth=5, rr_threshold=1.0, light_strategy=abi.LIGHTS_POWER, pixel_bounds=None):
    """CreateVolPathIntegrator (Enigma/VolPathIntegrator.cpp:179-200)."""
    d = abi.EnigmaVolPathDesc()
    sb = sample_bounds(film)
    d.max_depth = max_depth
    d.rr_threshold = rr_threshold
    d.light_strategy = light_strategy
    d.pixel_bounds[:] = sb if pixel_bounds is None else pixel_bounds
    d.samples_per_pixel = spp
    d.sample_bounds[:] = sb
    return d


def direct_lighting_desc(film, spp, max_depth=5, strategy=abi.DL_SAMPLE_ALL, pixel_bounds=None):
    """CreateDirectLightingIntegrator (Enigma/DirectLightingIntegrator.cpp:89-114)."""
    d = abi.EnigmaDirectLightingDesc()
    sb = sample_bounds(film)
    d.max_depth = max_depth
    d.strategy = strategy
    d.pixel_bounds[:] = sb if pixel_bounds is None else pixel_bounds
    d.samples_per_pixel = spp
    d.sample_bounds[:] = sb
    return d


# ----------------------------------------------------------------------------- mesh helpers
class MeshBuilder:
    def __init__(self):
        self.v, self.t, self.mat, self.light, self.flags = [], [], [], [], []
        self.materials, self.lights = [], []
        self.media, self.m_in, self.m_out = [], [], []
        self.interface = (-1, -1)   # MediumInterface (inside, outside) given to the triangles added next
        self.vn, self.vuv = [], []  # per-vertex normal / uv (zeros where the triangle has none)
        self.textures = []
        self.spheres = []

    def add_sphere(self, material, radius=1.0, object_to_world=None, emit=None, two_sided=False, n_samples=1, **kw):
        """emit: radiance of the sphere's DiffuseAreaLight (AreaLightSource "diffuse" in front of the shape)."""
        sp = abi.sphere(material, radius, object_to_world, **kw)
        if emit is not None:
            self.lights.append(abi.sphere_area_light(len(self.spheres), emit, two_sided, n_samples))
            sp.light = len(self.lights)
        self.spheres.append(sp)

    def add_texture(self, t):
        """Returns the value for a material's tex_* field (index + 1)."""
        self.textures.append(t)
        return len(self.textures)

    def add_medium(self, m):
        self.media.append(m)
        return len(self.media) - 1

    def medium_interface(self, inside=-1, outside=-1):
        self.interface = (inside, outside)

    def add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    @staticmethod
    def _ref_normal(p0, p1, p2):
        # Normalize(Cross(p0 - p2, p1 - p2)): the geometric normal Triangle::intersect assigns
        # (Enigma/TriangleMesh.cpp:287)
        return np.cross(np.asarray(p0, float) - p2, np.asarray(p1, float) - p2)

    def add_tri(self, p0, p1, p2, mat, toward=None, emit=None, two_sided=False, normals=None, uvs=None, reverse_orientation=False):
        """toward: a point the reference geometric normal must face (winding is swapped if not).
        normals / uvs: the three per-vertex normals / uvs of a mesh that has them (TriangleMesh::m_n, m_uv);
        reverse_orientation: the shape's ReverseOrientation flag (no handedness-swapping transform assumed)."""
        p0, p1, p2 = (np.asarray(p, dtype=np.float64) for p in (p0, p1, p2))
        if toward is not None:
            n = self._ref_normal(p0, p1, p2)
            if np.dot(n, np.asarray(toward, float) - p0) < 0:
                p1, p2 = p2, p1
                if normals is not None:
                    normals = (normals[0], normals[2], normals[1])
                if uvs is not None:
                    uvs = (uvs[0], uvs[2], uvs[1])
        base = len(self.v)
        self.v += [p0, p1, p2]
        self.vn += [np.zeros(3)] * 3 if normals is None else [np.asarray(n, dtype=np.float64) for n in normals]
        self.vuv += [np.zeros(2)] * 3 if uvs is None else [np.asarray(u, dtype=np.float64) for u in uvs]
        self.t.append((base, base + 1, base + 2))
        self.mat.append(mat)
        self.flags.append((abi.TRI_HAS_NORMALS if normals is not None else 0) | (abi.TRI_HAS_UVS if uvs is not None else 0) |
                          ((abi.TRI_REVERSE_ORIENTATION | abi.TRI_FLIP_NORMAL) if reverse_orientation else 0))
        self.m_in.append(self.interface[0])
        self.m_out.append(self.interface[1])
        if emit is not None:
            self.lights.append(abi.area_light(len(self.t) - 1, emit, two_sided))
            self.light.append(len(self.lights) - 1)
        else:
            self.light.append(-1)

    def add_quad(self, q, mat, toward=None, emit=None, two_sided=False, uvs=None):
        """uvs: one uv per corner (omitted: the reference's default per-triangle uvs)."""
        self.add_tri(q[0], q[1], q[2], mat, toward, emit, two_sided, uvs=None if uvs is None else (uvs[0], uvs[1], uvs[2]))
        self.add_tri(q[0], q[2], q[3], mat, toward, emit, two_sided, uvs=None if uvs is None else (uvs[0], uvs[2], uvs[3]))

    def add_point_light(self, pos, I):
        self.lights.append(abi.point_light(pos, I))

    def add_spot_light(self, pos, to, I, coneangle=30.0, conedeltaangle=5.0):
        self.lights.append(abi.spot_light(pos, to, I, coneangle, conedeltaangle))

    def add_distant_light(self, frm, to, L):
        self.lights.append(abi.distant_light(frm, to, L))

    def weld(self):
        """Share identical vertices: same position, normal and uv (keeps triangle order and winding)."""
        v = np.concatenate([np.asarray(self.v, dtype=np.float32), np.asarray(self.vn, dtype=np.float32), np.asarray(self.vuv, dtype=np.float32)], axis=1)
        uniq, inv = np.unique(v, axis=0, return_inverse=True)
        t = inv.reshape(-1)[np.asarray(self.t, dtype=np.int64)]
        return uniq, t.astype(np.uint32)

    def arrays(self, weld=True):
        if not self.t:      # spheres (or lights) only
            return abi.SceneArrays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), [], [], self.materials, self.lights,
                                   textures=self.textures, spheres=self.spheres, media=self.media)
        if weld:
            v, t = self.weld()
        else:
            v = np.concatenate([np.asarray(self.v, dtype=np.float32), np.asarray(self.vn, dtype=np.float32), np.asarray(self.vuv, dtype=np.float32)], axis=1)
            t = np.asarray(self.t, dtype=np.uint32)
        media = dict(media=self.media, tri_medium_inside=self.m_in, tri_medium_outside=self.m_out) if self.media else {}
        shading = {}
        if any(f & abi.TRI_HAS_NORMALS for f in self.flags):
            shading["normals"] = v[:, 3:6]
        if any(f & abi.TRI_HAS_UVS for f in self.flags):
            shading["uvs"] = v[:, 6:8]
        return abi.SceneArrays(v[:, :3], t, self.mat, self.light, self.materials, self.lights, tri_flags=self.flags, textures=self.textures, spheres=self.spheres, **media, **shading)


# ----------------------------------------------------------------------------- S1: Cornell box
_CORNELL = {
    "floor": [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)],
    "ceiling": [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)],
    "back": [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)],
    "right": [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)],
    "left": [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)],
    # the classic data puts the light in the ceiling plane (y = 548.8); it is lowered slightly
    # so no two surfaces are coplanar (closest-hit ties would depend on BVH traversal order).
    "light": [(343, 548.0, 227), (343, 548.0, 332), (213, 548.0, 332), (213, 548.0, 227)],
    "short": [
        [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
        [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
        [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
        [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
        [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)],
    ],
    "tall": [
        [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
        [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
        [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
        [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
        [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)],
    ],
}
CORNELL_CENTER = (278.0, 274.4, 279.6)
COPPER_ETA = (0.200, 0.924, 1.102)
COPPER_K = (3.912, 2.452, 2.142)


def cornell_box(variant="diffuse", delta_lights=False, extra_lights=()):
    """S1 (all matte) or S1m (short block metal, tall block rough glass).  32 triangles, 2 lights.
    delta_lights: adds a spot light aimed at the short block and a distant light shining in through the
    open front (its direction deliberately not unit length: the reference never normalises it)."""
    b = MeshBuilder()
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    red = b.add_material(abi.matte((0.65, 0.05, 0.05)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    walls = {}
    if variant == "diffuse":
        short_m = tall_m = white
    elif variant == "microfacet":
        short_m = b.add_material(abi.metal(COPPER_ETA, COPPER_K, roughness=0.1, remap=True))
        tall_m = b.add_material(abi.glass(1.5, 0.05, 0.05))
    elif variant == "specular":
        short_m = b.add_material(abi.plastic((0.25, 0.3, 0.6), (0.4, 0.4, 0.4), roughness=0.05))
        tall_m = b.add_material(abi.glass(1.5, 0.0, 0.0))
    elif variant == "uber":
        # the materials that are lists of lobes on the device: an uber block that is part diffuse, part glossy, part mirror,
        # part glass and 20 % see-through; a translucent block; a substrate floor
        short_m = b.add_material(abi.uber(kd=(0.25, 0.3, 0.6), ks=(0.3, 0.3, 0.3), kr=(0.2, 0.2, 0.2), kt=(0.3, 0.3, 0.3), opacity=(0.8, 0.8, 0.8),
                                          roughness=0.05, index=1.4))
        tall_m = b.add_material(abi.translucent(kd=(0.5, 0.45, 0.3), ks=(0.3, 0.3, 0.3), reflect=(0.5, 0.5, 0.5), transmit=(0.6, 0.5, 0.4), roughness=0.2))
        walls["floor"] = b.add_material(abi.substrate(kd=(0.5, 0.4, 0.3), ks=(0.1, 0.1, 0.1), uroughness=0.05, vroughness=0.2))
    elif variant == "mix":
        # mixes over every base material, a mix of a mix, and a mirror wall
        plastic = b.add_material(abi.plastic((0.25, 0.3, 0.6), (0.4, 0.4, 0.4), roughness=0.05))
        copper = b.add_material(abi.metal(COPPER_ETA, COPPER_K, roughness=0.1, remap=True))
        smooth = b.add_material(abi.glass(1.5, 0.0, 0.0))
        rough = b.add_material(abi.glass(1.4, 0.1, 0.05))
        short_m = b.add_material(abi.mix(plastic, copper, (0.4, 0.4, 0.4)))
        inner = b.add_material(abi.mix(smooth, rough, (0.5, 0.5, 0.5)))
        tall_m = b.add_material(abi.mix(inner, red, (0.7, 0.5, 0.9)))
        walls["back"] = b.add_material(abi.mirror((0.8, 0.85, 0.9)))
        walls["left"] = b.add_material(abi.mix(red, b.add_material(abi.matte((0.5, 0.5, 0.5), sigma=25.0)), (0.75, 0.75, 0.75)))
    else:
        raise ValueError(variant)
    c = CORNELL_CENTER
    for name, m in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        b.add_quad(_CORNELL[name], walls.get(name, m), toward=c)
    for name, m in (("short", short_m), ("tall", tall_m)):
        faces = _CORNELL[name]
        centre = np.mean([p for f in faces for p in f], axis=0)
        centre[1] = centre[1] * 0.5
        for f in faces:
            outward = np.mean(f, axis=0) * 2 - centre  # a point on the outer side of the face
            b.add_quad(f, m, toward=outward)
    b.add_quad(_CORNELL["light"], white, toward=(278, 0, 279.6), emit=(17.0, 12.0, 4.0))
    if delta_lights:
        b.add_spot_light((278, 500, 100), (186, 165, 169), (4.0e5, 5.0e5, 6.0e5), coneangle=25.0, conedeltaangle=10.0)
        b.add_distant_light((0.3, 0.5, -1.1), (0, 0, 0), (1.5, 1.2, 0.9))
    b.lights.extend(extra_lights)   # lights without a shape, after everything else
    return b.arrays()


def cornell_smooth(level=2):
    """The Cornell box with its blocks replaced by four coarse icospheres that exercise the shading geometry of
    Triangle::intersect (Enigma/TriangleMesh.cpp:238-262,290-371): an anisotropic metal sphere with per-vertex
    normals and spherical uvs (its tangent frame follows the uvs, seam included), a rough-glass sphere with normals
    only, a plastic sphere with uvs only, and a smooth-glass sphere with normals and ReverseOrientation."""
    b = MeshBuilder()
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    red = b.add_material(abi.matte((0.65, 0.05, 0.05)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    c = CORNELL_CENTER
    for name, m in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        b.add_quad(_CORNELL[name], m, toward=c)
    b.add_quad(_CORNELL["light"], white, toward=(278, 0, 279.6), emit=(17.0, 12.0, 4.0))
    sv, sf = _icosphere(level)
    uv = np.stack([np.arctan2(sv[:, 2], sv[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(sv[:, 1], -1, 1)) / math.pi], axis=1)
    spheres = (
        ((185.0, 90.0, 170.0), 90.0, b.add_material(abi.metal(COPPER_ETA, COPPER_K, remap=False, uroughness=0.04, vroughness=0.35)), True, True, False),
        ((390.0, 110.0, 340.0), 110.0, b.add_material(abi.glass(1.5, 0.05, 0.05)), True, False, False),
        ((400.0, 330.0, 330.0), 70.0, b.add_material(abi.plastic((0.25, 0.3, 0.6), (0.4, 0.4, 0.4), roughness=0.05)), False, True, False),
        ((150.0, 300.0, 300.0), 75.0, b.add_material(abi.glass(1.5, 0.0, 0.0)), True, False, True),
    )
    for centre, radius, mat, with_n, with_uv, reverse in spheres:
        centre = np.asarray(centre)
        for f in sf:
            p = sv[f] * radius + centre
            b.add_tri(p[0], p[1], p[2], mat, toward=p[0] + sv[f[0]], normals=tuple(sv[f]) if with_n else None,
                      uvs=tuple(uv[f]) if with_uv else None, reverse_orientation=reverse)
    return b.arrays()


def test_pattern(height=32, width=64, seed=0x7E87):
    """A small RGB image with every spatial frequency a MIP pyramid has to average: checker, gradient and noise."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:height, 0:width]
    checker = ((x // 4 + y // 4) % 2).astype(np.float32)
    img = np.stack([0.15 + 0.7 * checker, 0.1 + 0.8 * x / (width - 1.0), 0.2 + 0.6 * y / (height - 1.0)], -1).astype(np.float32)
    return (img * (0.75 + 0.25 * rng.rand(height, width, 1))).astype(np.float32)


def cornell_textured(trilinear=True, wrap=abi.WRAP_REPEAT):
    """The Cornell box with image textures on the material colours: the floor and the back wall matte with a tiled Kd
    (the back wall's uvs run past [0, 1] so the wrap mode shows), the short block plastic with textured Kd and Ks,
    the tall block rough glass with a textured Kt, the left wall a metal whose eta comes from a texture.  One of the
    textures is always made without "trilinear" (the reference then evaluates it to its average colour) and one with
    gamma false (every texel 1)."""
    b = MeshBuilder()
    img = test_pattern()
    tiled = b.add_texture(abi.image_texture(img, trilinear=trilinear, su=3.0, sv=2.0, du=0.25, wrap=wrap))
    plain = b.add_texture(abi.image_texture(img, trilinear=trilinear, wrap=wrap))
    average = b.add_texture(abi.image_texture(img * 0.6, trilinear=False))
    ones = b.add_texture(abi.image_texture(img, trilinear=trilinear, gamma=False, scale=0.9))
    eta_map = b.add_texture(abi.image_texture(0.2 + 1.5 * test_pattern(16, 16, seed=3), trilinear=trilinear))

    def mat(m, **tex):
        for k, v in tex.items():
            setattr(m, k, v)
        return b.add_material(m)
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    floor_m = mat(abi.matte((0.0, 0.0, 0.0)), tex_kd=tiled)
    back_m = mat(abi.matte((0.5, 0.5, 0.5)), tex_kd=plain)
    left_m = mat(abi.metal(COPPER_ETA, COPPER_K, roughness=0.2, remap=True), tex_eta=eta_map)
    short_m = mat(abi.plastic((0.25, 0.3, 0.6), (0.4, 0.4, 0.4), roughness=0.05), tex_kd=plain, tex_ks=average)
    tall_m = mat(abi.glass(1.5, 0.05, 0.05), tex_kt=ones)
    c = CORNELL_CENTER
    unit = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)]
    wide = [(-0.5, -0.25), (1.5, -0.25), (1.5, 1.25), (-0.5, 1.25)]
    b.add_quad(_CORNELL["floor"], floor_m, toward=c, uvs=unit)
    b.add_quad(_CORNELL["ceiling"], white, toward=c)
    b.add_quad(_CORNELL["back"], back_m, toward=c, uvs=wide)
    b.add_quad(_CORNELL["right"], green, toward=c)
    b.add_quad(_CORNELL["left"], left_m, toward=c, uvs=unit)
    for name, m in (("short", short_m), ("tall", tall_m)):
        faces = _CORNELL[name]
        centre = np.mean([p for f in faces for p in f], axis=0)
        centre[1] = centre[1] * 0.5
        for f in faces:
            b.add_quad(f, m, toward=np.mean(f, axis=0) * 2 - centre, uvs=unit if name == "short" else None)
    b.add_quad(_CORNELL["light"], white, toward=(278, 0, 279.6), emit=(17.0, 12.0, 4.0))
    return b.arrays()


def cornell_textured_lobes(trilinear=True):
    """The Cornell box with image textures on the materials that are lists of lobes: an uber floor whose Kd is a tiled
    texture and whose Ks a second one; a substrate back wall (Kd textured); a mirror left wall whose Kr comes from a texture
    made without "trilinear" (its average colour); a translucent short block whose Kd has BLACK texels -- where they
    are, the two Lambertian lobes do not exist (Enigma/Material.cpp:288-294) --; a tall block that mixes a textured matte
    with a smooth glass."""
    b = MeshBuilder()
    img = test_pattern()
    holes = test_pattern(16, 16, seed=11)
    holes[(np.indices((16, 16)).sum(axis=0) // 3) % 2 == 0] = 0.0
    tiled = b.add_texture(abi.image_texture(img, trilinear=trilinear, su=3.0, sv=2.0, du=0.25))
    plain = b.add_texture(abi.image_texture(img, trilinear=trilinear))
    average = b.add_texture(abi.image_texture(img * 0.9, trilinear=False))
    holed = b.add_texture(abi.image_texture(holes, trilinear=trilinear, su=2.0, sv=2.0))

    def mat(m, **tex):
        for k, v in tex.items():
            setattr(m, k, v)
        return b.add_material(m)
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    floor_m = mat(abi.uber((0.0, 0.0, 0.0), (0.3, 0.3, 0.3), kr=(0.1, 0.1, 0.1), opacity=(1.0, 0.9, 0.8), roughness=0.1), tex_kd=tiled, tex_ks=plain)
    back_m = mat(abi.substrate((0.5, 0.5, 0.5), (0.08, 0.08, 0.08), 0.1, 0.3), tex_kd=plain)
    left_m = mat(abi.mirror((0.9, 0.9, 0.9)), tex_kr=average)
    short_m = mat(abi.translucent((0.5, 0.5, 0.5), (0.0, 0.0, 0.0), reflect=(0.7, 0.6, 0.5), transmit=(0.3, 0.4, 0.5)), tex_kd=holed)
    matte_t = mat(abi.matte((0.5, 0.5, 0.5)), tex_kd=tiled)
    glass = b.add_material(abi.glass(1.5, 0.0, 0.0))
    tall_m = b.add_material(abi.mix(matte_t, glass, (0.6, 0.6, 0.6)))
    c = CORNELL_CENTER
    unit = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)]
    b.add_quad(_CORNELL["floor"], floor_m, toward=c, uvs=unit)
    b.add_quad(_CORNELL["ceiling"], white, toward=c)
    b.add_quad(_CORNELL["back"], back_m, toward=c, uvs=unit)
    b.add_quad(_CORNELL["right"], green, toward=c)
    b.add_quad(_CORNELL["left"], left_m, toward=c, uvs=unit)
    for name, m in (("short", short_m), ("tall", tall_m)):
        faces = _CORNELL[name]
        centre = np.mean([p for f in faces for p in f], axis=0)
        centre[1] = centre[1] * 0.5
        for f in faces:
            b.add_quad(f, m, toward=np.mean(f, axis=0) * 2 - centre, uvs=unit)
    b.add_quad(_CORNELL["light"], white, toward=(278, 0, 279.6), emit=(17.0, 12.0, 4.0))
    return b.arrays()


def _rotation(axis, degrees):
    a = np.asarray(axis, dtype=np.float64)
    a /= np.linalg.norm(a)
    th = math.radians(degrees)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    m = np.eye(4)
    m[:3, :3] = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)
    return m


def _translation(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def cornell_spheres(fog=False, textured=True, lamp=False):
    """The Cornell box with analytic Sphere shapes in place of the blocks: a matte sphere, a metal ellipsoid (rotated,
    non-uniformly scaled), a smooth-glass sphere, a rough-glass bowl (z range and phi sweep cut open, ReverseOrientation)
    and a plastic sphere under a mirroring transform.  fog: a scattering slab between camera and spheres and a tinted medium
    inside the glass sphere (VolPath).  textured: the matte sphere's Kd comes from a trilinear image texture over its (phi, theta)
    parametrisation.  lamp: a small emissive sphere (a DiffuseAreaLight on a Sphere, 2 samples) hangs under the ceiling."""
    b = MeshBuilder()
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    red = b.add_material(abi.matte((0.65, 0.05, 0.05)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    c = CORNELL_CENTER
    for name, m in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        b.add_quad(_CORNELL[name], m, toward=c)
    b.add_quad(_CORNELL["light"], white, toward=(278, 0, 279.6), emit=(17.0, 12.0, 4.0))
    if fog:
        med = b.add_medium(abi.homogeneous_medium((0.0005, 0.0005, 0.0005), (0.004, 0.005, 0.006), 0.2))
        lo, hi = np.array([40.0, 150.0, 20.0]), np.array([520.0, 300.0, 120.0])
        ctr = (lo + hi) / 2
        b.medium_interface(med, -1)
        for axis in range(3):
            for side, val in ((-1, lo[axis]), (1, hi[axis])):
                u, w = (axis + 1) % 3, (axis + 2) % 3
                q = []
                for du, dw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                    p = np.zeros(3)
                    p[axis] = val
                    p[u] = hi[u] if du else lo[u]
                    p[w] = hi[w] if dw else lo[w]
                    q.append(tuple(p))
                outward = ctr.copy()
                outward[axis] = val + side * 100.0
                b.add_quad(q, abi.NO_MATERIAL, toward=tuple(outward))
        b.medium_interface(-1, -1)
    globe = b.add_material(abi.matte((0.6, 0.6, 0.2), sigma=15.0))
    if textured:
        b.materials[globe].tex_kd = b.add_texture(abi.image_texture(test_pattern(), trilinear=True, su=2.0))   # u = phi / phimax, v over theta
    b.add_sphere(globe, radius=60.0, object_to_world=_translation(120, 60, 140) @ _rotation((1, 0, 0), -90.0))
    b.add_sphere(b.add_material(abi.metal(COPPER_ETA, COPPER_K, remap=False, uroughness=0.05, vroughness=0.25)), radius=1.0,
                 object_to_world=_translation(400, 90, 330) @ _rotation((1, 2, 0.5), 35.0) @ np.diag([110.0, 70.0, 90.0, 1.0]))
    tint = b.add_medium(abi.homogeneous_medium((0.004, 0.008, 0.016), (0.002, 0.002, 0.002), 0.0)) if fog else -1
    b.add_sphere(b.add_material(abi.glass(1.5, 0.0, 0.0)), radius=70.0, object_to_world=_translation(260, 70, 200), medium_inside=tint)
    b.add_sphere(b.add_material(abi.glass(1.5, 0.1, 0.1)), radius=80.0, z_min=-80.0, z_max=30.0, phi_max=300.0, reverse_orientation=True,
                 object_to_world=_translation(150, 330, 350) @ _rotation((1, 0, 0), -70.0))
    b.add_sphere(b.add_material(abi.plastic((0.25, 0.3, 0.6), (0.4, 0.4, 0.4), roughness=0.05)), radius=55.0,
                 object_to_world=_translation(420, 320, 300) @ np.diag([-1.0, 1.0, 1.0, 1.0]))
    if lamp:
        b.add_sphere(white, radius=25.0, object_to_world=_translation(330, 430, 200), emit=(40.0, 40.0, 60.0), n_samples=2)
    return b.arrays(weld=not fog)


def cornell_fog(variant="diffuse", g=0.3, extra_lights=()):
    """The Cornell box for the VolPathIntegrator: a slab of scattering fog (a box of material-less triangles whose
    MediumInterface has the fog inside) floats in front of the blocks, and the tall block -- glass in the
    microfacet / specular variants -- is filled with an absorbing, faintly scattering medium."""
    base = cornell_box(variant)
    b = MeshBuilder()
    for i in range(base.n_materials):
        b.add_material(base.materials[i])
    fog = b.add_medium(abi.homogeneous_medium((0.0005, 0.0005, 0.0005), (0.006, 0.007, 0.008), g))
    tint = b.add_medium(abi.homogeneous_medium((0.002, 0.004, 0.008), (0.001, 0.001, 0.001), 0.0))
    tall_first = 2 * 5 + 2 * 5                      # walls, short block, then the tall block's 10 triangles
    v = base.vertices[base.indices]
    for t in range(base.indices.shape[0]):
        b.medium_interface(tint, -1) if tall_first <= t < tall_first + 10 else b.medium_interface(-1, -1)
        emit = None if base.tri_light[t] < 0 else tuple(base.lights[base.tri_light[t]].L)
        b.add_tri(v[t, 0], v[t, 1], v[t, 2], int(base.tri_material[t]), emit=emit)
    lo, hi = np.array([60.0, 200.0, 40.0]), np.array([500.0, 420.0, 230.0])
    c = (lo + hi) / 2
    b.medium_interface(fog, -1)
    for axis in range(3):
        for side, val in ((-1, lo[axis]), (1, hi[axis])):
            u, w = (axis + 1) % 3, (axis + 2) % 3
            q = []
            for du, dw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = np.zeros(3)
                p[axis] = val
                p[u] = hi[u] if du else lo[u]
                p[w] = hi[w] if dw else lo[w]
                q.append(tuple(p))
            outward = c.copy()
            outward[axis] = val + side * 100.0
            b.add_quad(q, abi.NO_MATERIAL, toward=tuple(outward))
    b.lights.extend(extra_lights)
    return b.arrays(weld=False)


def sky_map(height=16, width=32):
    """A small latitude-longitude radiance map: a vertical gradient with a bright patch (a sun)."""
    a = np.zeros((height, width, 3), np.float32)
    th = (np.arange(height) + 0.5) / height
    a[:] = (np.stack([0.3 + 0.7 * (1 - th), 0.5 + 0.5 * (1 - th), 1.0 + 0 * th], -1)[:, None, :] * 0.5).astype(np.float32)
    a[height // 8:height // 4, width // 6:width // 4] = (30.0, 25.0, 15.0)
    return a


def cornell_sky(variant="microfacet", radiance_map=True):
    """The Cornell box (open towards the camera) with an InfiniteAreaLight added: a constant environment or the
    sky_map, rotated so that the map's pole points up (+y)."""
    base = cornell_box(variant)
    light = (abi.infinite_light((1.0, 1.0, 1.0), sky_map(), light_to_world=[[1, 0, 0], [0, 0, 1], [0, -1, 0]]) if radiance_map
             else abi.infinite_light((0.8, 0.9, 1.0)))
    lights = [base.lights[i] for i in range(base.n_lights)] + [light]
    return abi.SceneArrays(base.vertices, base.indices, base.tri_material, base.tri_light, [base.materials[i] for i in range(base.n_materials)], lights,
                           tri_flags=base.tri_flags)


def cornell_camera(xres, yres, lens_radius=0.0, focal_distance=1e30):
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))  # 39.3 degrees: the narrower image side
    return camera_desc((278, 273, -800), (278, 273, 0), (0, 1, 0), fov, xres, yres, lens_radius, focal_distance)


# ----------------------------------------------------------------------------- S3: many lights
def many_lights(n_quads=256, n_points=64, seed=0x11647):
    """Cornell geometry + n_quads small emissive quads on the ceiling/walls + point lights."""
    rng = np.random.RandomState(seed)
    b = MeshBuilder()
    white = b.add_material(abi.matte((0.73, 0.73, 0.73)))
    red = b.add_material(abi.matte((0.65, 0.05, 0.05)))
    green = b.add_material(abi.matte((0.12, 0.45, 0.15)))
    plast = b.add_material(abi.plastic((0.3, 0.3, 0.3), (0.5, 0.5, 0.5), roughness=0.08))
    c = CORNELL_CENTER
    for name, m in (("floor", plast), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        b.add_quad(_CORNELL[name], m, toward=c)
    for name in ("short", "tall"):
        faces = _CORNELL[name]
        centre = np.mean([p for f in faces for p in f], axis=0)
        centre[1] *= 0.5
        for f in faces:
            b.add_quad(f, white, toward=np.mean(f, axis=0) * 2 - centre)
    for _ in range(n_quads):
        x, z = rng.uniform(30, 520), rng.uniform(30, 520)
        y = rng.uniform(440, 540)
        s = rng.uniform(3, 9)
        rad = float(np.exp(rng.uniform(math.log(1.0), math.log(100.0))))
        col = rng.uniform(0.3, 1.0, size=3) * rad
        q = [(x - s, y, z - s), (x - s, y, z + s), (x + s, y, z + s), (x + s, y, z - s)]
        b.add_quad(q, white, toward=(x, 0, z), emit=tuple(float(v) for v in col))
    for _ in range(n_points):
        p = (rng.uniform(40, 510), rng.uniform(200, 430), rng.uniform(40, 510))
        inten = rng.uniform(2000, 20000) * rng.uniform(0.3, 1.0, size=3)
        b.add_point_light(p, tuple(float(v) for v in inten))
    return b.arrays()


# ----------------------------------------------------------------------------- S2: big mesh
def _icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
                  (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
                  (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    for _ in range(level):
        edges = {}
        verts = list(v)
        nf = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in edges:
                m = (verts[a] + verts[b]) * 0.5
                verts.append(m / np.linalg.norm(m))
                edges[key] = len(verts) - 1
            return edges[key]
        for a, b_, c in f:
            ab, bc, ca = mid(a, b_), mid(b_, c), mid(c, a)
            nf += [(a, ab, ca), (b_, bc, ab), (c, ca, bc), (ab, bc, ca)]
        v, f = np.array(verts), np.array(nf, dtype=np.int64)
    return v, f


def big_mesh(grid=512, sphere_level=6, n_spheres=8, seed=0xE16A):
    """S2: grid x grid displaced terrain (2*grid^2 triangles) + n_spheres icospheres
    (20*4^level triangles each) inside a closed box with one overhead quad light.
    Defaults: 524,288 + 8*81,920 + 12 + 2 = 1,179,662 triangles."""
    rng = np.random.RandomState(seed)
    mats = [abi.matte((0.55, 0.5, 0.45))]
    mats += [abi.metal(COPPER_ETA, COPPER_K, roughness=float(r), remap=True) for r in (0.05, 0.15, 0.3)]
    mats += [abi.glass(1.5, float(r), float(r)) for r in (0.0, 0.1, 0.2)]
    mats += [abi.matte((0.7, 0.7, 0.75))]
    size = 1000.0
    # terrain
    xs = np.linspace(0, size, grid + 1)
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    H = np.zeros_like(X)
    for _ in range(24):
        kx, kz = rng.uniform(0.5, 9, size=2) * 2 * math.pi / size
        H += rng.uniform(2, 14) * np.sin(kx * X + rng.uniform(0, 6.28)) * np.cos(kz * Z + rng.uniform(0, 6.28)) / 4
    H += rng.uniform(-0.3, 0.3, size=H.shape)
    tv = np.stack([X, H + 40.0, Z], axis=-1).reshape(-1, 3)
    i0 = (np.arange(grid)[:, None] * (grid + 1) + np.arange(grid)[None, :]).reshape(-1)
    # winding chosen so the reference normal cross(p0-p2, p1-p2) faces +y
    ta = np.stack([i0, i0 + 1, i0 + grid + 2], axis=1)
    tb = np.stack([i0, i0 + grid + 2, i0 + grid + 1], axis=1)
    tris = [np.stack([ta, tb], axis=1).reshape(-1, 3)]
    verts = [tv]
    tmat = [np.zeros(2 * grid * grid, dtype=np.uint32)]
    nv = tv.shape[0]
    sv, sf = _icosphere(sphere_level)
    for s in range(n_spheres):
        r = rng.uniform(60, 110)
        c = np.array([rng.uniform(150, 850), rng.uniform(170, 300), rng.uniform(150, 850)])
        verts.append(sv * r + c)
        # outward normal for cross(p0-p2, p1-p2): keep (a,b,c) if it faces out else swap
        tris.append(sf + nv)
        tmat.append(np.full(sf.shape[0], 1 + (s % 6), dtype=np.uint32))
        nv += sv.shape[0]
    V = np.concatenate(verts).astype(np.float32)
    T = np.concatenate(tris).astype(np.int64)
    M = np.concatenate(tmat)
    # fix winding of every non-terrain triangle to face away from its sphere centre / terrain up
    p0, p1, p2 = V[T[:, 0]].astype(np.float64), V[T[:, 1]].astype(np.float64), V[T[:, 2]].astype(np.float64)
    n = np.cross(p0 - p2, p1 - p2)
    want = np.zeros_like(n)
    want[: 2 * grid * grid] = (0, 1, 0)
    off = 2 * grid * grid
    k = 0
    for s in range(n_spheres):
        cnt = sf.shape[0]
        cen = verts[1 + s].mean(axis=0)
        want[off:off + cnt] = (p0[off:off + cnt] + p1[off:off + cnt] + p2[off:off + cnt]) / 3 - cen
        off += cnt
    swap = np.einsum("ij,ij->i", n, want) < 0
    T[swap, 1], T[swap, 2] = T[swap, 2].copy(), T[swap, 1].copy()
    b = MeshBuilder()
    b.materials = mats
    # enclosing box (inward facing) + light, appended after the bulk mesh
    box = [(-1, 0, -1), (size + 1, 0, -1), (size + 1, 0, size + 1), (-1, 0, size + 1)]
    top = [(x, 620.0, z) for x, _, z in box]
    quads = [box, top, [box[0], box[1], top[1], top[0]], [box[1], box[2], top[2], top[1]], [box[2], box[3], top[3], top[2]],
             [box[3], box[0], top[0], top[3]]]
    centre = (size / 2, 300.0, size / 2)
    for q in quads:
        b.add_quad(q, 7, toward=centre)
    b.add_quad([(350, 600, 350), (350, 600, 650), (650, 600, 650), (650, 600, 350)], 7, toward=centre, emit=(40.0, 38.0, 34.0))
    ev = np.asarray(b.v, dtype=np.float32)
    et = np.asarray(b.t, dtype=np.int64) + V.shape[0]
    V = np.concatenate([V, ev])
    T = np.concatenate([T, et]).astype(np.uint32)
    M = np.concatenate([M, np.asarray(b.mat, dtype=np.uint32)])
    Lt = np.concatenate([np.full(T.shape[0] - len(b.t), -1, dtype=np.int32), np.asarray(b.light, dtype=np.int32)])
    lights = b.lights
    for l in lights:
        l.triangle += T.shape[0] - len(b.t)
    return abi.SceneArrays(V, T, M, Lt, mats, lights)


def big_mesh_fog(grid=512, sphere_level=6, n_spheres=8, seed=0xE16A):
    """The defined part of the VolPathIntegrator configuration: the big_mesh scene as a TEXTURED mesh inside a homogeneous
    medium.  The terrain carries uvs (16 x 16 tiles) and a trilinear Kd image texture; the spheres carry spherical uvs, one
    kind of them is a plastic with trilinear Kd and Ks textures, one a smooth glass filled with a tinted absorbing medium.
    A box of material-less triangles with scattering fog inside (MediumInterface fog | none) encloses terrain air space and
    spheres; camera and light sit outside it, so every camera ray and every shadow ray crosses a medium boundary (camera
    rays carry no medium of their own in the reference, Enigma/Camera.cpp:144-192).  Same triangle count as big_mesh + 12."""
    base = big_mesh(grid, sphere_level, n_spheres, seed)
    V, T = base.vertices, base.indices
    nt_terrain = 2 * grid * grid
    nv_terrain = (grid + 1) * (grid + 1)
    sv, sf = _icosphere(sphere_level)
    nt_bulk = nt_terrain + n_spheres * sf.shape[0]
    nv_bulk = nv_terrain + n_spheres * sv.shape[0]
    size = 1000.0
    uv = np.zeros((V.shape[0], 2), dtype=np.float32)
    uv[:nv_terrain, 0] = V[:nv_terrain, 0] / size * 16.0
    uv[:nv_terrain, 1] = V[:nv_terrain, 2] / size * 16.0
    suv = np.stack([np.arctan2(sv[:, 2], sv[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(sv[:, 1], -1, 1)) / math.pi], axis=1)
    for s_ in range(n_spheres):
        a = nv_terrain + s_ * sv.shape[0]
        uv[a:a + sv.shape[0]] = suv * (4.0, 2.0)
    flags = np.zeros(T.shape[0], dtype=np.uint32)
    flags[:nt_bulk] = abi.TRI_HAS_UVS
    img = test_pattern(64, 64, seed=0xF06)
    textures = [abi.image_texture(img, trilinear=True), abi.image_texture(0.25 + 0.6 * test_pattern(32, 32, seed=5), trilinear=True, su=2.0, sv=2.0),
                abi.image_texture(0.2 + 0.5 * test_pattern(16, 32, seed=9), trilinear=True)]
    mats = [base.materials[i] for i in range(base.n_materials)]
    mats[0] = abi.matte((0.55, 0.5, 0.45)); mats[0].tex_kd = 1
    mats[4] = abi.plastic((0.3, 0.3, 0.3), (0.4, 0.4, 0.4), roughness=0.08); mats[4].tex_kd = 2; mats[4].tex_ks = 3
    mats[5] = abi.glass(1.5, 0.1, 0.1)
    mats[6] = abi.glass(1.5, 0.0, 0.0)
    fog = abi.homogeneous_medium((0.0001, 0.0001, 0.0001), (0.0012, 0.0013, 0.0015), 0.3)
    tint = abi.homogeneous_medium((0.002, 0.004, 0.008), (0.0005, 0.0005, 0.0005), 0.0)
    m_in = np.full(T.shape[0], -1, dtype=np.int32)
    m_out = np.full(T.shape[0], -1, dtype=np.int32)
    glass_filled = base.tri_material == 6      # the smooth-glass spheres: tinted medium inside, the fog outside
    m_in[glass_filled] = 1
    m_out[glass_filled] = 0
    # the fog box: material-less triangles, fog inside
    b = MeshBuilder()
    lo, hi = np.array([20.0, 60.0, 35.0]), np.array([980.0, 450.0, 980.0])
    c = (lo + hi) / 2
    for axis in range(3):
        for side, val in ((-1, lo[axis]), (1, hi[axis])):
            u, w = (axis + 1) % 3, (axis + 2) % 3
            q = []
            for du, dw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = np.zeros(3)
                p[axis] = val
                p[u] = hi[u] if du else lo[u]
                p[w] = hi[w] if dw else lo[w]
                q.append(tuple(p))
            outward = c.copy()
            outward[axis] = val + side * 100.0
            b.add_quad(q, abi.NO_MATERIAL, toward=tuple(outward))
    bv = np.asarray(b.v, dtype=np.float32)
    bt = np.asarray(b.t, dtype=np.uint32) + V.shape[0]
    nb = bt.shape[0]
    V2 = np.concatenate([V, bv])
    T2 = np.concatenate([T, bt])
    uv2 = np.concatenate([uv, np.zeros((bv.shape[0], 2), dtype=np.float32)])
    M2 = np.concatenate([base.tri_material, np.full(nb, abi.NO_MATERIAL, dtype=np.uint32)])
    L2 = np.concatenate([base.tri_light, np.full(nb, -1, dtype=np.int32)])
    F2 = np.concatenate([flags, np.zeros(nb, dtype=np.uint32)])
    in2 = np.concatenate([m_in, np.zeros(nb, dtype=np.int32)])
    out2 = np.concatenate([m_out, np.full(nb, -1, dtype=np.int32)])
    lights = [base.lights[i] for i in range(base.n_lights)]
    return abi.SceneArrays(V2, T2, M2, L2, mats, lights, tri_flags=F2, media=[fog, tint], tri_medium_inside=in2, tri_medium_outside=out2,
                           uvs=uv2, textures=textures)


def big_mesh_camera(xres, yres):
    return camera_desc((500, 330, 20), (500, 140, 560), (0, 1, 0), 50.0, xres, yres)
