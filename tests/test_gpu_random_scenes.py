"""Differential test on random scenes: the GPU library against the CPU oracle, seed by seed (tools/fuzz_scenes.py).

A seed draws a soup of triangles (degenerate, tiny and huge ones among them, some with per-vertex normals / uvs),
materials of every kind, emissive triangles, delta lights, sometimes spheres, media and an environment light, a camera,
one of the three integrators with random depth, Russian-roulette threshold, light strategy and pixel filter.  The two
sides must trace the same rays (counts within 1 %: a libm-level difference can flip a branch of a rough-glass path) and
produce the same image (mean relative error below 1 %, median pixel difference below 1e-3 relative); a description that
the library refuses must be one the oracle flags as undefined in the reference, or one of the two documented refusals
the library decides from the scene alone.  4500 seeds of tools/fuzz_scenes.py ran on the MI355X without a mismatch; a
fixed slice of them runs here."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_scenes  # noqa: E402

# 241: throughput at the edge of the float range; 1656, 2704: an emissive sliver with vertex normals, whose single-precision
# normal points the other way than its exact one; 19, 23: emissive triangles of no area with vertex normals
SEEDS = list(range(0, 48)) + [241, 1656, 2704]


@pytest.mark.gpu
def test_random_scenes_render_like_the_oracle():
    statuses = {}
    exact = 0
    for seed in SEEDS:
        res = fuzz_scenes.compare(seed)
        statuses[seed] = res["status"]
        assert res["status"] != "mismatch", res
        if res["status"] == "ok":
            assert res["finite"] and res["stack_overflows"] == 0, res
            exact += res["counts_apart"] == 0
    n_ok = sum(s == "ok" for s in statuses.values())
    assert n_ok >= len(SEEDS) * 3 // 4, statuses          # refusals are the exception
    assert exact >= n_ok * 3 // 4, (exact, n_ok)           # and most scenes agree to the ray


@pytest.mark.gpu
def test_random_scene_files_render_like_the_oracle(tmp_path):
    """The same scenes as pbrt-format text through raytrace_amd/enigma_render (scene-description front end, host
    mirror classes, C ABI, GPU) against the oracle rendering the scene built in Python: 1200 seeds ran without a
    mismatch; 580 is a portrait film, whose screen window the reference forms in single precision
    (Enigma/Camera.cpp:385-432)."""
    statuses = {}
    for seed in list(range(100, 124)) + [580]:
        res = fuzz_scenes.compare_file(seed, str(tmp_path))
        statuses[seed] = res["status"]
        assert res["status"] != "mismatch", res
    assert sum(s == "ok" for s in statuses.values()) >= len(statuses) * 2 // 3, statuses


def test_random_scene_generator_covers_the_feature_space():
    """No GPU: the seeds used above reach every integrator, spheres, media, every pixel filter, and the oracle
    renders a finite image for the first few."""
    import orc
    labels = [fuzz_scenes.random_case(seed)[4] for seed in SEEDS]
    for word in ("path", "direct-one", "direct-all", "volpath"):
        assert any(l.startswith(word + " ") for l in labels), word
    for word in ("filter box", "filter triangle", "filter gaussian", "filter mitchell", "filter sinc", "media 1", "strategy 0", "strategy 1", "strategy 2"):
        assert any(word in l for l in labels), word
    assert any(" spheres 0 " not in l for l in labels)
    rendered = 0
    for seed in SEEDS[:8]:
        sc, cam, film, integ, _ = fuzz_scenes.random_case(seed)
        o = orc.Oracle(sc, cam, film, integ)
        try:
            o.render(0, 4, threads=4)
        except RuntimeError as e:  # a combination the reference leaves undefined, or one the oracle does not restate
            assert "undefined" in str(e) or "not restated" in str(e), e
        else:
            assert np.isfinite(o.image()).all()
            rendered += 1
        o.close()
    assert rendered >= 5
