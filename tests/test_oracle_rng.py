"""CPU: the oracle's RNG restatement (shaders/random.glsl:27-64).

Parity status: the reference holds no vectors for these functions, so they are checked (a) against
an independent pure-Python restatement written from the GLSL text, (b) for the exactness properties
the GLSL implies, (c) against the committed oracle_kat.npz so the oracle cannot drift silently."""
import math
import os

import numpy as np

from conftest import GOLDEN, bits

KAT = np.load(os.path.join(GOLDEN, "oracle_kat.npz"))
M32 = 0xFFFFFFFF


def py_tea(val0, val1):
    v0, v1, s0 = val0, val1, 0
    for _ in range(16):
        s0 = (s0 + 0x9E3779B9) & M32
        v0 = (v0 + ((((v1 << 4) & M32) + 0xA341316C & M32) ^ ((v1 + s0) & M32) ^ (((v1 >> 5) + 0xC8013EA4) & M32))) & M32
        v1 = (v1 + ((((v0 << 4) & M32) + 0xAD90777D & M32) ^ ((v0 + s0) & M32) ^ (((v0 >> 5) + 0x7E95761E) & M32))) & M32
    return v0


def py_lcg(prev):
    prev = (1664525 * prev + 1013904223) & M32
    return prev, prev & 0x00FFFFFF


def test_tea_matches_python_restatement_and_kat(O):
    for a, b, want in zip(KAT["tea_v0"], KAT["tea_v1"], KAT["tea_out"]):
        assert O.tea(int(a), int(b)) == py_tea(int(a), int(b)) == int(want)
    rng = np.random.default_rng(1)
    for a, b in rng.integers(0, 2 ** 32, size=(200, 2)):
        assert O.tea(int(a), int(b)) == py_tea(int(a), int(b))


def test_lcg_stream_and_rnd_exact(O):
    for s0, want in zip(KAT["lcg_state0"], KAT["lcg_out"]):
        got, final = O.lcg_stream(int(s0), 16)
        s = int(s0)
        exp = []
        for _ in range(16):
            s, o = py_lcg(s)
            exp.append(o)
        assert got == exp == [int(x) for x in want]
        assert final == s
        # rnd = lcg / 2^24 exactly (random.glsl:54-57)
        r, _ = O.rnd_stream(int(s0), 16)
        assert np.array_equal(r.astype(np.float64) * 16777216.0, np.array(exp, dtype=np.float64))
        assert r.min() >= 0.0 and r.max() < 1.0


def test_sincos_definition(O):
    # committed values
    for k, want in zip(KAT["sincos_k"], KAT["sincos_out"]):
        s, c = O.sincos_2pi(np.float32(k) / np.float32(16777216.0))
        assert np.array_equal(bits([s, c]), bits(want))
    # accuracy vs the real functions of the exact angle, and vs what the GLSL text computes
    # (theta = rnd*2.0*M_PI in fp32, then sin/cos): both within the precision GLSL guarantees (2^-11).
    worst_exact = worst_glsl = 0.0
    for k in list(range(0, 1 << 24, 65521)) + [1, 2 ** 22, 2 ** 23, 3 * 2 ** 22, 2 ** 24 - 1]:
        u = np.float32(k) / np.float32(16777216.0)
        s, c = O.sincos_2pi(u)
        a = 2.0 * math.pi * (k / 16777216.0)
        worst_exact = max(worst_exact, abs(float(s) - math.sin(a)), abs(float(c) - math.cos(a)))
        theta = np.float32(np.float32(u * np.float32(2.0)) * np.float32(3.14159265))
        worst_glsl = max(worst_glsl, abs(float(s) - math.sin(float(theta))), abs(float(c) - math.cos(float(theta))))
        assert abs(float(s) * float(s) + float(c) * float(c) - 1.0) < 4e-7
    assert worst_exact < 1.2e-7
    assert worst_glsl < 1e-6 < 2.0 ** -11


def test_unit_vector(O):
    for s0, want in zip(KAT["lcg_state0"], KAT["unit_out"]):
        v, final = O.rnd_unit_vector(int(s0))
        assert np.array_equal(bits(v), bits(want))
        # two lcg steps consumed, theta first then z (random.glsl:60-61)
        st, o1 = py_lcg(int(s0))
        st, o2 = py_lcg(st)
        assert final == st
        assert v[2] == np.float32(np.float32(o2 / 16777216.0) * np.float32(2.0) - np.float32(1.0))
        assert abs(float(np.dot(v.astype(np.float64), v.astype(np.float64))) - 1.0) < 1e-6


def test_sample_seed_quirk(O):
    """raytrace.rgen:34-35,58: the raygen-local seed advances by the two jitter draws only; the
    payload seed is a copy, so sample i+1's jitter draws equal sample i's first bounce draws."""
    s0 = O.tea(123, 4)
    r, _ = O.rnd_stream(s0, 8)
    jitter = [(r[2 * i], r[2 * i + 1]) for i in range(3)]
    # bounce stream of sample 0 starts after its two jitter draws
    _, st = O.rnd_stream(s0, 2)
    b, _ = O.rnd_stream(st, 2)
    assert (b[0], b[1]) == jitter[1]
