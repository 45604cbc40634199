"""The arithmetic the opt-in split-bf16 GEMM rests on (csrc/gemm_bf16x3.hip), restated in numpy on the CPU: the three bf16
pieces of an f32 value, and the six piece products that stand for an f32 product."""
import numpy as np


def _trunc16(x):  # the upper 16 bits of an f32 = a bf16 value, as f32
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _bf16_rne(x):  # round to nearest even to bf16, as f32 (finite inputs)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """a0 = trunc16(a), a1 = trunc16(a - a0), a2 = bf16(a - a0 - a1): what split4() in the kernel computes."""
    a0 = _trunc16(x)
    r = (x - a0).astype(np.float32)
    a1 = _trunc16(r)
    a2 = _bf16_rne((r - a1).astype(np.float32))
    return a0, a1, a2


def _values(n, seed):
    g = np.random.default_rng(seed)
    x = (g.standard_normal(n) * np.power(10.0, g.uniform(-12, 12, n))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38, 65504.0]
    return x


def test_three_pieces_restore_the_value_exactly():
    x = _values(200000, 0)
    a0, a1, a2 = split3(x)
    # every piece is a bf16 value (low 16 bits clear) and the subtractions in split3 were exact: the f64 sum is the value
    for p in (a0, a1, a2):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    assert np.array_equal(a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64), x.astype(np.float64))
    # piece magnitudes fall by 2^-8 per level (the weights the six-term selection relies on)
    nz = np.abs(x) > 1e-30
    assert np.all(np.abs(a1[nz]) <= np.abs(x[nz]) * 2.0 ** -7)
    assert np.all(np.abs(a2[nz]) <= np.abs(x[nz]) * 2.0 ** -15)


def test_six_piece_products_are_an_f32_equivalent_product():
    a, b = _values(200000, 1), _values(200000, 2)[::-1].copy()
    keep = (np.abs(a) > 1e-15) & (np.abs(b) > 1e-15) & (np.abs(a) < 1e15) & (np.abs(b) < 1e15)
    a, b = a[keep], b[keep]
    A, B = split3(a), split3(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = sum(A[i].astype(np.float64) * B[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))
    # the three dropped products (a1 b2, a2 b1, a2 b2): a truncated piece can be as large as 2^-7 of what it was cut from, so the
    # worst single product is off by 2^-21.4 (median 2^-25); in a dot product these errors are signed and average out — the GPU
    # tests measure the kernel's error against fp64 at or below the f32-MFMA kernel's (tests/test_gemm_bf16x3_gpu.py)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -21 and np.median(rel[rel > 0]) <= 2.0 ** -24
    # and each kept product of two bf16 values is exact in f32 (8 + 8 significand bits)
    p = (A[0] * B[1]).astype(np.float32)
    assert np.array_equal(p.astype(np.float64), A[0].astype(np.float64) * B[1].astype(np.float64))
