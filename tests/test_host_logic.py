"""CPU tests of the product's host logic: C ABI surface, Sobol' generator + pixel-table factorisation,
BVH builder invariants, scene builders, film description."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import common
import orc
import raytrace_amd as ra
from raytrace_amd import abi, scenes, distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def hostcheck():
    out = os.path.join(ROOT, "tests", "_hostcheck.so")
    srcs = [os.path.join(ROOT, "tests", "hostcheck.cpp"), os.path.join(ROOT, "raytrace_amd", "csrc", "bvh_build.cpp"),
            os.path.join(ROOT, "raytrace_amd", "csrc", "sobol_host.cpp")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", out] + srcs)
    L = C.CDLL(out)
    L.hc_index.restype = C.c_uint64
    L.hc_index.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    L.hc_bits.restype = C.c_uint32
    L.hc_bits.argtypes = [C.c_uint64, C.c_int]
    return L


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed for that) and exports what include/enigma_hip.h declares."""
    L = ra.load_library()
    hdr = open(os.path.join(ROOT, "include", "enigma_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(enigma_[a-z_]+)\s*\(", hdr))
    assert declared == set(ra.SYMBOLS), declared ^ set(ra.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None


def test_ctypes_structs_match_the_header():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "enigma_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(EnigmaMaterial), sizeof(EnigmaLight), sizeof(EnigmaSceneDesc), sizeof(EnigmaCameraDesc),
         sizeof(EnigmaFilmDesc), sizeof(EnigmaIntegratorDesc), sizeof(EnigmaStats));
  printf("%zu %zu %zu %zu\n", offsetof(EnigmaSceneDesc, materials), offsetof(EnigmaIntegratorDesc, samples_per_pixel),
         offsetof(EnigmaMaterial, index), offsetof(EnigmaFilmDesc, max_sample_luminance));
  return 0; }'''
    exe = "/tmp/enigma_abi_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = subprocess.check_output([exe]).decode().split()
    want = [C.sizeof(abi.EnigmaMaterial), C.sizeof(abi.EnigmaLight), C.sizeof(abi.EnigmaSceneDesc), C.sizeof(abi.EnigmaCameraDesc),
            C.sizeof(abi.EnigmaFilmDesc), C.sizeof(abi.EnigmaIntegratorDesc), C.sizeof(abi.EnigmaStats),
            abi.EnigmaSceneDesc.materials.offset, abi.EnigmaIntegratorDesc.samples_per_pixel.offset, abi.EnigmaMaterial.index.offset,
            abi.EnigmaFilmDesc.max_sample_luminance.offset]
    assert [int(v) for v in out] == want


def test_calls_without_gpu_fail_loudly():
    """On a box without a GPU the library must refuse, not fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ra.EnigmaError):
        ra.Renderer(0)


def test_cpp_sobol_generator_matches_reference_digests(hostcheck):
    d = json.load(open(os.path.join(GOLD, "sobol_digest.json")))
    m32 = np.zeros(1024 * 52, dtype=np.uint32)
    vdc = np.zeros(25 * 52, dtype=np.uint64)
    vdci = np.zeros(26 * 52, dtype=np.uint64)
    hostcheck.hc_sobol_tables(abi.ptr(m32), abi.ptr(vdc), abi.ptr(vdci))
    assert hashlib.sha256(m32.astype("<u4").tobytes()).hexdigest() == d["sobol32_sha256"]
    assert hashlib.sha256(vdc.astype("<u8").tobytes()).hexdigest() == d["vdc_sha256"]
    assert hashlib.sha256(vdci.astype("<u8").tobytes()).hexdigest() == d["vdc_inv_sha256"]


@pytest.mark.parametrize("m", [0, 1, 5, 9, 11])
def test_pixel_table_factorisation_is_exact(hostcheck, m):
    """Tx[px] ^ Ty[py] ^ c(sampleNum) == SobolSample(SobolIntervalToIndex(m, sampleNum, p), d), all bits."""
    rng = np.random.RandomState(m)
    n, nd = 400, 45
    res = 1 << m
    px = rng.randint(0, res, n).astype(np.uint32)
    py = rng.randint(0, res, n).astype(np.uint32)
    sn = rng.randint(0, 1024, n).astype(np.uint64)
    out = np.zeros((n, nd), dtype=np.uint32)
    hostcheck.hc_factored(m, nd, n, abi.ptr(px), abi.ptr(py), abi.ptr(sn), abi.ptr(out))
    for i in range(0, n, 7):
        idx = hostcheck.hc_index(m, int(sn[i]), int(px[i]), int(py[i]))
        for d in (0, 1, 2, 5, 17, 44):
            assert int(out[i, d]) == hostcheck.hc_bits(idx, d)
    # and against the oracle's bit loops on the numpy-generated tables
    if m == 9:
        sc, cam, film, integ = common.cornell_setup(res=512, spp=1024)
        o = orc.Oracle(sc, cam, film, integ)
        idx_o, dims_o = o.sobol(np.stack([px, py], axis=1).astype(np.int32), sn.astype(np.int64), 8)
        for i in range(n):
            assert int(idx_o[i]) == hostcheck.hc_index(m, int(sn[i]), int(px[i]), int(py[i]))
        want = np.minimum(out[:, 2:8].astype(np.float32) * np.float32(2.0 ** -32), np.float32(0.9999999403953552))
        assert np.array_equal(want, dims_o[:, 2:8])
        o.close()


@pytest.mark.parametrize("builder,kw", [("cornell", {}), ("many", {}), ("big", {"grid": 48, "sphere_level": 3, "n_spheres": 3})])
def test_bvh_builder_invariants(hostcheck, builder, kw):
    sc = {"cornell": scenes.cornell_box, "many": scenes.many_lights, "big": scenes.big_mesh}[builder](**kw)
    nT = sc.indices.shape[0]
    rng = np.random.default_rng(5)
    lo = rng.uniform(-100, 600, (17, 3)).astype(np.float32)
    boxes = np.concatenate([lo, lo + rng.uniform(0, 80, (17, 3)).astype(np.float32)], axis=1)   # primitives that are not triangles
    for nB in (0, 17):
        order = np.zeros(nT + nB, dtype=np.uint32)
        checks = np.zeros(5, dtype=np.uint32)
        nodes = hostcheck.hc_bvh(abi.ptr(sc.vertices), abi.ptr(sc.indices), nT, abi.ptr(boxes), nB, abi.ptr(order), abi.ptr(checks))
        assert sorted(order.tolist()) == list(range(nT + nB))       # a permutation
        assert checks[1] == nT + nB and checks[4] == 0              # every primitive in exactly one leaf
        assert checks[2] == 0 and checks[3] == 0                    # stored boxes hold what is below them; encoding consistent
        assert 1 <= checks[0] <= 64 and nodes >= 1


def _bvh_checks(hostcheck, v, idx, boxes=None):
    nT = idx.shape[0]
    nB = 0 if boxes is None else boxes.shape[0]
    order = np.zeros(max(1, nT + nB), dtype=np.uint32)
    checks = np.zeros(5, dtype=np.uint32)
    b = boxes if nB else np.zeros((1, 6), dtype=np.float32)
    hostcheck.hc_bvh(abi.ptr(v), abi.ptr(idx), nT, abi.ptr(b), nB, abi.ptr(order), abi.ptr(checks))
    return checks


def test_bvh_builder_degenerate_inputs(hostcheck):
    # all triangles identical (coincident centroids), a single triangle, none at all
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    for nT in (0, 1, 2, 37):
        idx = np.tile(np.array([0, 1, 2], dtype=np.uint32), (max(nT, 1), 1))[:nT].reshape(nT, 3)
        checks = _bvh_checks(hostcheck, v, idx)
        assert checks[1] == nT and checks[2] == 0 and checks[3] == 0 and checks[4] == 0
    # boxes only; a grid of coplanar quads (flat nodes on one axis)
    boxes = np.array([[0, 0, 0, 1, 1, 1], [5, 5, 5, 6, 7, 8]], dtype=np.float32)
    checks = _bvh_checks(hostcheck, v, np.zeros((0, 3), dtype=np.uint32), boxes)
    assert checks[1] == 2 and checks[2] == 0 and checks[3] == 0
    g = 24
    xs, zs = np.meshgrid(np.arange(g + 1, dtype=np.float32), np.arange(g + 1, dtype=np.float32))
    vv = np.stack([xs.ravel(), np.full(xs.size, 5, dtype=np.float32), zs.ravel()], axis=1)
    q = np.array([[i * (g + 1) + j, i * (g + 1) + j + 1, (i + 1) * (g + 1) + j] for i in range(g) for j in range(g)], dtype=np.uint32)
    checks = _bvh_checks(hostcheck, vv, q)
    assert checks[1] == g * g and checks[2] == 0 and checks[3] == 0 and checks[4] == 0


def test_bvh_depth_is_bounded_for_adversarial_geometry(hostcheck):
    """Triangles whose sizes grow geometrically around one corner make surface-area splits peel one triangle per level;
    the builder stops splitting by area at a fixed depth and halves by count, so the tree stays within the 64 levels the
    traversal stacks hold."""
    n = 400
    s = (1.07 ** np.arange(n)).astype(np.float32)
    v = np.zeros((3 * n, 3), dtype=np.float32)
    v[1::3, 0] = s; v[2::3, 1] = s; v[0::3, 2] = s * 1e-3
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    checks = _bvh_checks(hostcheck, v, idx)
    assert checks[1] == n and checks[2] == 0 and checks[3] == 0 and checks[4] == 0
    assert checks[0] <= 64


def test_bvh_builder_extreme_coordinates(hostcheck):
    """Coordinates out to 1e18 (the library accepts up to 2^60): surface areas reach 1e37 and their sums over a tree
    leave single precision, so the cost model runs in double; every triangle must still be found in a leaf and every
    stored box must hold what lies below it.  Denormal coordinates likewise.  Non-finite input, which the library
    refuses before it gets here, must at least not take the process down (bins are clamped)."""
    rng = np.random.RandomState(69)
    for scale, n in ((1e14, 2000), (1e17, 300), (1e-40, 300)):
        v = (rng.normal(size=(3 * n, 3)) * 10 ** rng.uniform(-3, 3) * scale).astype(np.float32)
        v = np.clip(v, -1.1e18, 1.1e18)
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        checks = _bvh_checks(hostcheck, v, idx)
        assert checks[1] == n and checks[2] == 0 and checks[3] == 0 and checks[4] == 0 and checks[0] <= 64, (scale, checks)
    for bad in (np.nan, np.inf, -np.inf):
        v = rng.normal(size=(150, 3)).astype(np.float32)
        v[rng.randint(150, size=5), rng.randint(3, size=5)] = bad
        _bvh_checks(hostcheck, v, np.arange(150, dtype=np.uint32).reshape(50, 3))


def test_bvh_builder_on_several_threads_builds_the_same_tree(hostcheck, monkeypatch):
    """Scenes of 2^15 primitives and more are built on several threads: the large ranges in chunks, the subtrees below
    them by workers, the collapse and the emission side by side.  The result -- primitive order and node count -- must not
    depend on the number of threads, and the invariants must hold."""
    sc = scenes.big_mesh(grid=192, sphere_level=5, n_spheres=4)      # 155,648 triangles
    nT = sc.indices.shape[0]
    assert nT > 1 << 17
    results = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("ENIGMA_BVH_THREADS", threads)
        order = np.zeros(nT, dtype=np.uint32)
        checks = np.zeros(5, dtype=np.uint32)
        nodes = hostcheck.hc_bvh(abi.ptr(sc.vertices), abi.ptr(sc.indices), nT, abi.ptr(np.zeros((1, 6), dtype=np.float32)), 0, abi.ptr(order), abi.ptr(checks))
        assert checks[1] == nT and checks[2] == 0 and checks[3] == 0 and checks[4] == 0 and checks[0] <= 64
        results.append((nodes, checks[0], order.copy()))
    for nodes, depth, order in results[1:]:
        assert nodes == results[0][0] and depth == results[0][1] and np.array_equal(order, results[0][2])


def test_scene_builders():
    c = scenes.cornell_box("diffuse")
    assert c.indices.shape[0] == 32 and c.n_lights == 2
    # the light's reference normal cross(p0 - p2, p1 - p2) faces down, walls face the box centre
    v, t = c.vertices.astype(np.float64), c.indices
    n = np.cross(v[t[:, 0]] - v[t[:, 2]], v[t[:, 1]] - v[t[:, 2]])
    lights = np.where(c.tri_light >= 0)[0]
    assert np.all(n[lights, 1] < 0)
    centre = np.array(scenes.CORNELL_CENTER)
    walls = np.arange(10)
    assert np.all(np.einsum("ij,ij->i", n[walls], centre - v[t[walls, 0]]) > 0)
    m = scenes.many_lights()
    assert m.n_lights == 2 * 256 + 64 and m.indices.shape[0] == 30 + 512
    b = scenes.big_mesh(grid=32, sphere_level=2, n_spheres=2)
    assert b.indices.shape[0] == 2 * 32 * 32 + 2 * 320 + 14
    assert b.indices.max() < b.vertices.shape[0]


def test_film_desc_and_sample_bounds():
    f = scenes.film_desc(64, 48, "box")
    assert list(f.cropped_pixel_bounds) == [0, 0, 64, 48] and all(v == 1.0 for v in f.filter_table)
    assert scenes.sample_bounds(f) == [0, 0, 64, 48]
    g = scenes.film_desc(64, 48, "gaussian", crop=(0.25, 0.75, 0.5, 1.0))
    assert list(g.cropped_pixel_bounds) == [16, 24, 48, 48]
    assert scenes.sample_bounds(g) == [14, 22, 50, 50]            # grown by radius 2 - 0.5
    tab = np.array(g.filter_table[:]).reshape(16, 16)
    assert tab[0, 0] == tab.max() and tab[15, 15] >= 0 and np.all(np.diff(tab[0]) <= 0)


def test_shard_range_partitions():
    for b, e, w in ((0, 1024, 8), (0, 16, 3), (5, 6, 4), (0, 0, 2)):
        parts = [distributed.shard_range(b, e, r, w) for r in range(w)]
        assert parts[0][0] == b and parts[-1][1] == e
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))


def test_recorded_bench_line_keeps_the_contract():
    """The last default bench line kept under profiles/ carries every key the driver reads, the roofline object is
    consistent with itself, and the figures read from a counter pass are labelled with the commit of that pass."""
    import glob
    import json
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*_bench_default.json")))
    assert lines
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "Mrays/s" and d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for name in ("roofline", "roofline_shadow", "roofline_shade"):
        r = d[name]
        assert r["bound"] in ("hbm", "valu") and r["unit"] in ("GB/s", "Tlane-op/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1.5
        assert r["avg_launch_ms"] > 0 and r["launches"] > 0
        if r.get("recorded"):
            assert r["recorded"].get("commit") and r["recorded"].get("source")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    b = d["bigmesh"]
    assert b["job_measured"] is True and b["job_spp"] == 1024 and b["ms_bvh_build"] > 0 and b["reduce"].startswith(("none", "native"))
