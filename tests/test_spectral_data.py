"""The spectral tables behind the default metal and the prime bases of the radical inverse, pinned to the reference's
own tables by sha256 (tests/golden/spectral_digest.json, written by tools/derive_spectral_data.py from the reference's
text), and the two restatements of RGBSpectrum::FromSampled (oracle, product host code) against each other and against
closed forms.  No GPU."""
import hashlib
import json
import os

import numpy as np

import orc
import raytrace_amd as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGEST = json.load(open(os.path.join(ROOT, "tests", "golden", "spectral_digest.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _table(name):
    rows = [l.split() for l in open(os.path.join(ROOT, "raytrace_amd", "data", name)) if not l.startswith("#")]
    return np.array([[float(t) for t in r] for r in rows], dtype=np.float64)


def test_data_files_and_compiled_tables_match_the_reference_digests():
    cie, cu = _table("cie_xyz_1nm.txt"), _table("copper_nk.txt")
    assert cie.shape == (471, 4) and cu.shape == (56, 3)
    for col, name in enumerate(("CIE_lambda", "CIE_X", "CIE_Y", "CIE_Z")):
        assert _sha(cie[:, col].astype("<f4")) == DIGEST[name], name
    for col, name in enumerate(("CopperWavelengths", "CopperN", "CopperK")):
        assert _sha(cu[:, col].astype("<f4")) == DIGEST[name], name
    # what the oracle (and, through the same generated include, the product) was compiled with
    ocie, ocu = orc.spectral_tables()
    for row, name in enumerate(("CIE_lambda", "CIE_X", "CIE_Y", "CIE_Z")):
        assert _sha(ocie[row].astype("<f4")) == DIGEST[name], name
    for row, name in enumerate(("CopperWavelengths", "CopperN", "CopperK")):
        assert _sha(ocu[row].astype("<f4")) == DIGEST[name], name
    assert np.array_equal(cie[:, 0], np.arange(360, 831))


def test_prime_table_of_the_radical_inverse_is_the_first_thousand_primes():
    """RadicalInverse(baseIndex, a) uses base Primes[baseIndex] (Enigma/SamplerInclude.h:44-62,2397-2560); the spatial light
    distribution takes bases 2, 3, 5, 7, 11 = Primes[0..4].  A sieve reproduces the reference's table digest."""
    sieve = np.ones(8000, dtype=bool); sieve[:2] = False
    for i in range(2, 90):
        if sieve[i]:
            sieve[i * i::i] = False
    primes = np.nonzero(sieve)[0][:1000].astype("<i4")
    assert primes.shape[0] == 1000 and list(primes[:5]) == [2, 3, 5, 7, 11]
    assert _sha(primes) == DIGEST["Primes"]
    sums = np.concatenate([[0], np.cumsum(primes)[:-1]]).astype("<i4")
    assert _sha(sums) == DIGEST["PrimeSums"]


def test_from_sampled_closed_forms():
    # a constant spectrum c has XYZ = c * (sum of the matching functions) * scale; Y comes out as c * sum(Y) * 470 / (106.856895 * 471)
    cie = _table("cie_xyz_1nm.txt")
    f = np.float32
    one = orc.rgb_from_sampled([300.0, 900.0], [1.0, 1.0])
    xyz = [np.float64(cie[:, c].sum()) * 470.0 / (106.856895 * 471) for c in (1, 2, 3)]
    want = [3.240479 * xyz[0] - 1.537150 * xyz[1] - 0.498535 * xyz[2], -0.969256 * xyz[0] + 1.875991 * xyz[1] + 0.041556 * xyz[2],
            0.055648 * xyz[0] - 0.204043 * xyz[1] + 1.057311 * xyz[2]]
    assert np.allclose(one, want, rtol=2e-5)
    assert abs(xyz[1] - 1.0) < 3e-3                                  # the Y integral normalises a flat spectrum's luminance to 1
    assert np.allclose(orc.rgb_from_sampled([300.0, 900.0], [2.5, 2.5]), 2.5 * one, rtol=1e-5)   # linear in the values (up to float summation)
    # unsorted samples are sorted first: same answer
    lam, v = np.array([400, 500, 600, 700], dtype=f), np.array([0.2, 0.9, 0.4, 0.7], dtype=f)
    perm = [2, 0, 3, 1]
    assert np.array_equal(orc.rgb_from_sampled(lam, v), orc.rgb_from_sampled(lam[perm], v[perm]))
    # a spectrum that is non-zero only around 450 nm is blue, around 620 nm red
    blue = orc.rgb_from_sampled([300, 440, 450, 460, 900], [0, 0, 1, 0, 0])
    red = orc.rgb_from_sampled([300, 610, 620, 630, 900], [0, 0, 1, 0, 0])
    assert blue[2] > blue[0] and blue[2] > blue[1] and red[0] > red[1] and red[0] > red[2]


def test_default_metal_is_copper_in_both_restatements():
    eta_o, k_o = orc.default_metal()
    eta_p, k_p = ra.default_metal_spectra()
    assert np.array_equal(eta_o.view(np.uint32), eta_p.view(np.uint32)) and np.array_equal(k_o.view(np.uint32), k_p.view(np.uint32))
    # close to the RGB that pbrt's own (SampledSpectrum) route gives for the same data: (0.2004, 0.9240, 1.1022), (3.9129, 2.4528, 2.1421);
    # RGBSpectrum::FromSampled sums 471 samples against a 470 nm range, which lowers every value by 1 / 471
    assert np.allclose(eta_o, np.array((0.2004, 0.9240, 1.1022)) * 470 / 471, atol=1e-3)
    assert np.allclose(k_o, np.array((3.9129, 2.4528, 2.1421)) * 470 / 471, atol=2e-3)
    # an independent float64 evaluation of the same integral from the data files
    cie, cu = _table("cie_xyz_1nm.txt"), _table("copper_nk.txt")
    for col, got in ((1, eta_o), (2, k_o)):
        val = np.interp(cie[:, 0], cu[:, 0], cu[:, col])
        xyz = [(val * cie[:, c]).sum() * 470.0 / (106.856895 * 471) for c in (1, 2, 3)]
        want = [3.240479 * xyz[0] - 1.537150 * xyz[1] - 0.498535 * xyz[2], -0.969256 * xyz[0] + 1.875991 * xyz[1] + 0.041556 * xyz[2],
                0.055648 * xyz[0] - 0.204043 * xyz[1] + 1.057311 * xyz[2]]
        assert np.allclose(got, want, rtol=1e-4)
