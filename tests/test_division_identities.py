"""Two fp32 divisions of the reference run on the device as a multiplication in double followed by one rounding:
  * rgb2grey's  (r + g + b) / 3.f            -> (float)((double)s * (1.0 / 3.0))        (pyramid.hip, third())
  * the descriptor's  x / hist_w              -> (float)((double)x * (1.0 / (double)hist_w))   (descriptor.hip)
Both are claimed to BE the correctly rounded fp32 quotient (the double product is within 2^-52 of the true quotient,
and a quotient of two fp32 numbers is never that close to an fp32 rounding boundary).  Checked here against IEEE fp32
division (numpy) on every mantissa for the constant divisor and on tens of millions of operand pairs for the general one."""
import numpy as np


def test_division_by_three_every_mantissa():
    m = np.arange(1 << 23, dtype=np.uint32)
    inv3 = np.float64(1.0) / np.float64(3.0)
    for exp in (0, 1, 2, 7, 9, -7, -30, 60):          # grey sums live in [0, 765]; a few far exponents for good measure
        bits = ((np.uint32(127 + exp) << np.uint32(23)) | m).astype(np.uint32)
        s = bits.view(np.float32)
        want = s / np.float32(3.0)
        got = (s.astype(np.float64) * inv3).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), exp
    for v in (0.0, 765.0, 1.0, 3.0, 2.9999998):
        s = np.float32(v)
        assert (s / np.float32(3.0)).tobytes() == np.float32(np.float64(s) * inv3).tobytes()


def test_quotient_through_a_double_reciprocal():
    rng = np.random.default_rng(11)
    for rep in range(8):
        n = 4_000_000
        x = (rng.standard_normal(n) * np.ldexp(1.0, rng.integers(-12, 13, n))).astype(np.float32)
        hw = (rng.uniform(1.0, 2.0, n) * np.ldexp(1.0, rng.integers(-3, 8, n))).astype(np.float32)     # hist_w = 3 * scale factor: a few .. a few hundred
        want = x / hw
        got = (x.astype(np.float64) * (np.float64(1.0) / hw.astype(np.float64))).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), rep
    # structured operands: small integers over small integers (exact and repeating quotients), equal operands, powers of two
    a = np.arange(-300, 301, dtype=np.float32)[:, None]
    b = np.arange(1, 400, dtype=np.float32)[None, :]
    want = (a / b).astype(np.float32)
    got = (a.astype(np.float64) * (np.float64(1.0) / b.astype(np.float64))).astype(np.float32)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
