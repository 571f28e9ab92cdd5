"""CPU model of the matcher's ranking keys (openpano_amd/csrc/match.hip): two-term bf16 split of both operands, the three
bf16 x bf16 products per element accumulated in fp32 from -|y|^2 / 2, four low mantissa bits given to the slot index.
DESIGN.md section 1.8 bounds the distance of such a key from the true score x.y - |y|^2 / 2 by e = 3.3e-5 (|x|^2 + |y|^2)
and builds the re-score margin E = 8.2e-5 (|x|^2 + max |y|^2) on it.  This test measures that distance on RootSIFT-like
and on adversarial descriptor sets (near-duplicates, sparse, saturated) in numpy -- the bound is about the arithmetic,
not about the hardware; what the hardware does with it is checked by the whole-job GPU tests."""
import numpy as np


def _bf16(v):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(v, np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def _keys(x, y, order):
    xh = _bf16(x); xl = _bf16(x - xh)
    yh = _bf16(y); yl = _bf16(y - yh)
    ny = np.zeros(len(y), np.float32)
    for k in range(128):
        ny = (ny + y[:, k] * y[:, k]).astype(np.float32)
    acc = (-(ny * np.float32(0.5))).astype(np.float32)
    for k in order:                                    # any order: the bound does not depend on it
        for a, b in ((xh, yh), (xh, yl), (xl, yh)):
            acc = (acc + (a[:, k] * b[:, k]).astype(np.float32)).astype(np.float32)      # bf16 x bf16 is exact in fp32
    key = (acc.view(np.uint32) & np.uint32(0xFFFFFFF0)).view(np.float32)                # slot bits
    return key


def _rootsift(rng, n, sparsity):
    h = rng.gamma(0.6, 1.0, (n, 128)) * (rng.random((n, 128)) > sparsity)
    h[:, 0] += 1e-3
    return (np.sqrt(h / h.sum(axis=1, keepdims=True)) * 512).astype(np.float32)


def test_key_error_stays_inside_the_budget():
    rng = np.random.default_rng(5)
    worst = 0.0
    for sparsity in (0.0, 0.5, 0.9):
        x = _rootsift(rng, 4000, sparsity); y = _rootsift(rng, 4000, sparsity)
        cases = [(x, y), (x, x.copy()), (x, (x * np.float32(1.0009765625)).astype(np.float32)), (x, y[::-1].copy())]
        for a, b in cases:
            for order in (range(128), range(127, -1, -1), rng.permutation(128)):
                key = _keys(a, b, order).astype(np.float64)
                a64, b64 = a.astype(np.float64), b.astype(np.float64)
                true = (a64 * b64).sum(axis=1) - 0.5 * (b64 * b64).sum(axis=1)
                scale = (a64 * a64).sum(axis=1) + (b64 * b64).sum(axis=1)
                worst = max(worst, float(np.max(np.abs(key - true) / scale)))
    assert worst <= 3.3e-5, worst
    assert worst > 1e-7                                  # the model really rounds
