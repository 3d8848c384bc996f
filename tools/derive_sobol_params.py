#!/usr/bin/env python3
"""Dev-time tool: recover the Joe-Kuo generating parameters (degree s, polynomial
coefficient word a, initial direction integers m_1..m_s) of every Sobol' dimension
from a table of 52-bit generator-matrix columns, and write them as a compact text
table.  The product and the oracle regenerate the matrices from that table with
the published Joe-Kuo recurrence; nothing of the table's text is kept.

Input: the reference's tabulated matrices, read as text (study only):
  /root/reference/Enigma/SobolMatrices.cpp  (SobolMatrices32/64, VdCSobolMatrices[Inv])
Outputs:
  raytrace_amd/data/joe_kuo_1024.txt    "s a m_1 ... m_s" per dimension >= 2
  tests/golden/sobol_digest.json        sha256 digests of the four tables so the
                                        regenerated ones can be pinned without the
                                        reference being present.
"""
import hashlib, json, re, sys
import numpy as np

REF = "/root/reference/Enigma/SobolMatrices.cpp"
NDIM, NCOL = 1024, 52

def parse_tables(path):
    txt = open(path).read()
    out = {}
    for name in ("SobolMatrices32", "SobolMatrices64", "VdCSobolMatrices", "VdCSobolMatricesInv"):
        m = re.search(r"\b%s\s*\[[^\]]*\]\s*(\[[^\]]*\])?\s*=\s*\{" % name, txt)
        assert m, name
        depth, i = 1, m.end()
        start = i
        while depth:
            c = txt[i]
            depth += (c == "{") - (c == "}")
            i += 1
        body = re.sub(r"//[^\n]*", "", txt[start:i - 1])
        if name.startswith("VdC"):
            rows = re.findall(r"\{([^{}]*)\}", body)
            out[name] = [[int(t.rstrip("ULu"), 16) for t in re.findall(r"0x[0-9a-fA-F]+(?:ULL)?", r)] for r in rows]
        else:
            out[name] = [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", body)]
    return out

def derive(cols52):
    """cols52: 52 ints V_1..V_52 with V_k = m_k << (52-k). Return (s, a, m[1..s])."""
    m = [cols52[k] >> (NCOL - 1 - k) for k in range(NCOL)]
    assert all((m[k] << (NCOL - 1 - k)) == cols52[k] for k in range(NCOL))
    for s in range(1, 20):
        for a in range(1 << max(s - 1, 0)):
            ok = True
            for k in range(s, NCOL):
                v = m[k - s] ^ (m[k - s] << s)
                for i in range(1, s):
                    if (a >> (s - 1 - i)) & 1:
                        v ^= m[k - i] << i
                if v != m[k]:
                    ok = False
                    break
            if ok:
                return s, a, m[:s]
    raise RuntimeError("no recurrence found")

def main():
    t = parse_tables(REF)
    m64 = np.array(t["SobolMatrices64"], dtype=np.uint64).reshape(NDIM, NCOL)
    m32 = np.array(t["SobolMatrices32"], dtype=np.uint32).reshape(NDIM, NCOL)
    assert np.array_equal((m64 >> np.uint64(20)).astype(np.uint32), m32)
    lines = []
    # dimension 1 (index 0) is van der Corput: m_k = 1
    assert all(int(m64[0, k]) == 1 << (NCOL - 1 - k) for k in range(NCOL))
    for d in range(1, NDIM):
        s, a, mi = derive([int(x) for x in m64[d]])
        lines.append("%d %d %s" % (s, a, " ".join(str(x) for x in mi)))
    with open("raytrace_amd/data/joe_kuo_1024.txt", "w") as f:
        f.write("# Joe-Kuo Sobol' parameters for dimensions 2..1024: s a m_1..m_s\n")
        f.write("\n".join(lines) + "\n")
    vdc = np.array([r + [0] * (NCOL - len(r)) for r in t["VdCSobolMatrices"]], dtype=np.uint64)
    vdci = np.array([r + [0] * (NCOL - len(r)) for r in t["VdCSobolMatricesInv"]], dtype=np.uint64)
    dig = {
        "sobol32_sha256": hashlib.sha256(m32.astype("<u4").tobytes()).hexdigest(),
        "sobol64_sha256": hashlib.sha256(m64.astype("<u8").tobytes()).hexdigest(),
        "vdc_rows": int(vdc.shape[0]), "vdc_inv_rows": int(vdci.shape[0]),
        "vdc_row_lens": [len(r) for r in t["VdCSobolMatrices"]],
        "vdc_inv_row_lens": [len(r) for r in t["VdCSobolMatricesInv"]],
        "vdc_sha256": hashlib.sha256(vdc.astype("<u8").tobytes()).hexdigest(),
        "vdc_inv_sha256": hashlib.sha256(vdci.astype("<u8").tobytes()).hexdigest(),
        "layout": "little-endian, row-major [dim][52] / [m-1][52] zero padded",
    }
    json.dump(dig, open("tests/golden/sobol_digest.json", "w"), indent=1)
    print(dig)

if __name__ == "__main__":
    main()
