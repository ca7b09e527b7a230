"""Direction numbers of the Sobol' sampler (reference src/samplers/sobol.cpp, sobolseq.{h,cpp}).

The reference tabulates the generator matrices of S. Joe and F. Y. Kuo, "Constructing Sobol sequences with better two-dimensional
projections", SIAM J. Sci. Comput. 30 (2008) -- data file new-joe-kuo-6.21201, first 1024 dimensions -- as 1024 x 52 machine
words (Matrices::matrices32 / matrices64) plus the matrices its pixel look-up needs.  None of that is code.  This script reads
the words AS DATA from the reference checkout and writes
  * ray-tracer_amd/csrc/kkd_sobol_dirnums.h: the PUBLISHED form of the same data, one line per dimension as in Joe and Kuo's
    file (degree s, polynomial coefficients a, initial direction numbers m_1 ... m_s), recovered from the matrix columns
    (m_k = column k >> (32 - k); s and a by matching the defining recurrence), from which the library regenerates the matrices
    with its own code (kkd_host.cpp: sobolMatrices).  The script regenerates all 1024 x 52 words from the recovered numbers and
    refuses to write anything unless they equal the reference's words.
  * tests/golden/sobol_kat.json: known answers -- a checksum of the full 32-bit table, and values of sobol::sampleSingle and
    sobol::look_up (sobolseq.h:43-60, 102-137, transcribed below in Python and evaluated on the reference's own tables) for random
    arguments -- against which library and oracle are tested bit for bit.
Run once, in the container that holds /root/reference:   python scripts/make_sobol_dirnums.py"""
import json
import os
import re
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/samplers/sobolseq.cpp"
DIMS, SIZE = 1024, 52


def parse_tables(text):
    def block(name):
        i = text.index(name)
        i = text.index("=", i)
        j = text.index("\n};", i)
        return text[i:j]

    def words(t):
        return [int(w.rstrip("UL"), 16) for w in re.findall(r"0x[0-9a-fA-F]+U?L*", t)]

    def rows(name):
        """rows of at most 52 words ('{ // m = k'), padded with zeros to 52 -> flat list indexed [(m - 1) * 52 + c]"""
        flat = []
        for row in re.findall(r"\{\s*// m = \d+(.*?)\}", block(name), re.S):
            w = words(row)
            assert len(w) <= SIZE
            flat += w + [0] * (SIZE - len(w))
        return flat
    m32 = words(block("Matrices::matrices32["))
    vdc, inv = rows("Matrices::vdc_sobol_matrices[][52]"), rows("Matrices::vdc_sobol_matrices_inv[][52]")
    assert len(m32) == DIMS * SIZE and min(len(vdc), len(inv)) >= 16 * SIZE, (len(m32), len(vdc), len(inv))
    return m32, vdc, inv


def columns52(s, a, m):
    """Joe & Kuo's construction, 52 bits: v_k = m_k << (52 - k); m_k for k > s by the recurrence of the primitive polynomial"""
    m = list(m)
    for k in range(s, SIZE):
        x = m[k - s] ^ (m[k - s] << s)
        for i in range(1, s):
            if (a >> (s - 1 - i)) & 1:
                x ^= m[k - i] << i
        m.append(x)
    return [m[k] << (SIZE - 1 - k) for k in range(SIZE)]


def recover(words32):
    """(s, a, m_1..m_s) of one dimension from the top 32 bits of its 52 columns"""
    m = [words32[k] >> (31 - k) for k in range(32)]           # m_{k+1}: k+1 bits
    for s in range(1, 20):
        for a in range(1 << max(0, s - 1)):
            cols = columns52(s, a, m[:s])
            if all((cols[k] >> 20) == words32[k] for k in range(SIZE)):
                return s, a, m[:s]
    raise SystemExit("no primitive polynomial of degree < 20 reproduces a dimension")


def sample_single(m32, index, dim, scramble=0):               # sobolseq.h:43-60
    r, i = scramble, dim * SIZE
    while index:
        if index & 1:
            r ^= m32[i]
        index >>= 1; i += 1
    return r                                                   # x 2^-32, capped below 1, by the caller


def look_up(vdc, inv, m, frame, px, py, scramble=0):          # sobolseq.h:102-137 (single precision)
    index = frame << (2 * m)
    delta, c = 0, 0
    while frame:
        if frame & 1:
            delta ^= vdc[(m - 1) * SIZE + c]
        frame >>= 1; c += 1
    scramble = (scramble & 0xFFFFFFFF) >> (32 - m)
    b = (((px ^ scramble) << m) | (py ^ scramble)) ^ delta
    c = 0
    while b:
        if b & 1:
            index ^= inv[(m - 1) * SIZE + c]
        b >>= 1; c += 1
    return index


if __name__ == "__main__":
    import random
    m32, vdc, inv = parse_tables(open(REF).read())
    dims = []
    # dimension 0 is the van der Corput sequence (identity matrix): not part of Joe and Kuo's file
    assert all(m32[k] == (1 << (31 - k) if k < 32 else 0) for k in range(SIZE))
    for d in range(1, DIMS):
        dims.append(recover(m32[d * SIZE:(d + 1) * SIZE]))
    regenerated = [(1 << (31 - k) if k < 32 else 0) for k in range(SIZE)]
    for s, a, m in dims:
        regenerated += [c >> 20 for c in columns52(s, a, m)]
    assert regenerated == m32, "regenerated matrices differ from the reference's"
    crc = zlib.crc32(b"".join(w.to_bytes(4, "little") for w in m32)) & 0xFFFFFFFF
    out = os.path.join(ROOT, "ray-tracer_amd", "csrc", "kkd_sobol_dirnums.h")
    with open(out, "w") as f:
        f.write("/* kkd_sobol_dirnums.h -- GENERATED by scripts/make_sobol_dirnums.py: direction numbers of S. Joe and F. Y. Kuo (new-joe-kuo-6.21201),\n"
                " * dimensions 2 ... 1024 in their numbering (dimension 1 is the van der Corput sequence), the data the reference tabulates as\n"
                " * generator matrices in src/samplers/sobolseq.cpp.  Per dimension: degree s, coefficients a of the primitive polynomial, then\n"
                " * m_1 ... m_s.  kkd_host.cpp: sobolMatrices regenerates the 1024 x 52 matrix words; their CRC-32 must be kSobolMatrixCrc. */\n"
                "#pragma once\n#include <cstdint>\nnamespace kkd {\n")
        f.write("static constexpr uint32_t kSobolDims = %d, kSobolSize = %d, kSobolMatrixCrc = 0x%08xu;\n" % (DIMS, SIZE, crc))
        flat = []
        for s, a, m in dims:
            flat += [s, a] + m
        f.write("static const uint32_t kJoeKuo[%d] = {\n" % len(flat))
        pos = 0
        for s, a, m in dims:
            f.write("    " + ", ".join(str(v) for v in [s, a] + m) + ",\n")
        f.write("};\n} /* namespace kkd */\n")
    rng = random.Random(2012)
    kat = {"source": "scripts/make_sobol_dirnums.py: sobol::sampleSingle / sobol::look_up of src/samplers/sobolseq.h evaluated on the tables of src/samplers/sobolseq.cpp",
           "matrices32_crc32": crc, "sample": [], "look_up": [], "first_points": []}
    for i in range(16):
        kat["first_points"].append([sample_single(m32, i, d) for d in range(8)])
    for _ in range(400):
        index, dim = rng.getrandbits(rng.choice([8, 20, 36, 44])), rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 13, 100, 511, 1023])
        kat["sample"].append([index, dim, sample_single(m32, index, dim)])
    for _ in range(400):
        m = rng.randint(2, 13)
        frame, px, py = rng.getrandbits(rng.choice([1, 4, 10, 14])), rng.randrange(1 << m), rng.randrange(1 << m)
        kat["look_up"].append([m, frame, px, py, look_up(vdc, inv, m, frame, px, py)])
    json.dump(kat, open(os.path.join(ROOT, "tests", "golden", "sobol_kat.json"), "w"))
    print("dimensions", len(dims) + 1, "max degree", max(d[0] for d in dims), "numbers", sum(2 + d[0] for d in dims), "crc 0x%08x" % crc)
