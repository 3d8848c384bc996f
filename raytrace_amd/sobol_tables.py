"""Sobol' generator matrices from the Joe-Kuo parameter table (numpy, host side).

Regenerates what the reference tabulates in Enigma/SobolMatrices.cpp:6 (SobolMatrices32),
:26637 (VdCSobolMatrices) and :26763 (VdCSobolMatricesInv) from the public Joe-Kuo
parameters (new-joe-kuo-6.21201; Enigma/SobolMatrices.cpp:2-3 names the source):

* 52-bit direction numbers V_k = m_k << (52-k) by the Joe-Kuo recurrence; the 32-bit
  matrices are their top 32 bits.
* for a 2^m x 2^m pixel grid, the first two dimensions' top m bits are a GF(2)-linear
  map of the low 2m index bits (matrix A) plus a map of the remaining "frame" bits
  (matrix B).  VdCSobolMatrices[m-1][c] is column c of B and VdCSobolMatricesInv[m-1]
  holds the columns of A^-1, which is what SobolIntervalToIndex
  (Enigma/SamplerInclude.h:2600-2620) consumes.

This module feeds the test oracle; the product library has its own C++ generator.
"""
import os
import numpy as np

NUM_DIMS = 1024
MATRIX_SIZE = 52
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "joe_kuo_1024.txt")


def direction_numbers_52(path=_DATA):
    """[1024][52] python ints, column k of dimension d as a 52-bit fixed point number."""
    out = [[1 << (MATRIX_SIZE - 1 - k) for k in range(MATRIX_SIZE)]]
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip() and not ln.startswith("#")]
    assert len(rows) == NUM_DIMS - 1
    for r in rows:
        s, a = int(r[0]), int(r[1])
        m = [int(x) for x in r[2:2 + s]]
        assert len(m) == s
        for k in range(s, MATRIX_SIZE):
            v = m[k - s] ^ (m[k - s] << s)
            for i in range(1, s):
                if (a >> (s - 1 - i)) & 1:
                    v ^= m[k - i] << i
            m.append(v)
        out.append([m[k] << (MATRIX_SIZE - 1 - k) for k in range(MATRIX_SIZE)])
    return out


def _gf2_inverse(cols, n):
    """cols[j] = column j (n-bit int) of an n x n GF(2) matrix; returns columns of the inverse."""
    # Gauss-Jordan on rows: build row bitmasks of [A | I]
    rows = []
    for i in range(n):
        r = 0
        for j in range(n):
            if (cols[j] >> i) & 1:
                r |= 1 << j
        rows.append(r | (1 << (n + i)))
    for c in range(n):
        piv = next(i for i in range(c, n) if (rows[i] >> c) & 1)
        rows[c], rows[piv] = rows[piv], rows[c]
        for i in range(n):
            if i != c and (rows[i] >> c) & 1:
                rows[i] ^= rows[c]
    inv_rows = [r >> n for r in rows]
    inv_cols = []
    for j in range(n):
        v = 0
        for i in range(n):
            if (inv_rows[i] >> j) & 1:
                v |= 1 << i
        inv_cols.append(v)
    return inv_cols


def generate():
    """Returns (sobol32 [1024*52] u32, vdc [25*52] u64, vdc_inv [26*52] u64), zero padded rows."""
    v52 = direction_numbers_52()
    m32 = np.array([[c >> 20 for c in row] for row in v52], dtype=np.uint32)
    d0 = [int(x) for x in m32[0]]
    d1 = [int(x) for x in m32[1]]
    vdc = np.zeros((25, MATRIX_SIZE), dtype=np.uint64)
    vdc_inv = np.zeros((26, MATRIX_SIZE), dtype=np.uint64)
    for m in range(1, 27):
        def pix(col):
            return ((d0[col] >> (32 - m)) << m) | (d1[col] >> (32 - m))
        a_cols = [pix(j) for j in range(2 * m)]
        inv = _gf2_inverse(a_cols, 2 * m)
        vdc_inv[m - 1, :2 * m] = inv
        if m <= 25:
            ncols = MATRIX_SIZE - 2 * m
            vdc[m - 1, :ncols] = [pix(2 * m + c) for c in range(ncols)]
    return m32.reshape(-1), vdc.reshape(-1), vdc_inv.reshape(-1)
