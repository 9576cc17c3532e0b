"""Directed device test of the FP64 FFT arithmetic (csrc/fft_dev.hpp): negacyclic products of limb polynomials through the
transforms, the prepared-operand scaling and the multiply-accumulate exactly as the fused kernels call them, returned RAW
(before the rounding the path applies) and compared with exact integer arithmetic on the host.

The path is exact as long as the FP64 round-off of an accumulated product stays below 1/2 (the contract of the reference's
own FFT64 backend, /root/reference/examples/fhe-ram.rs:3-7).  The end-to-end parity tests only ever see operands of
typical magnitude; these drive the documented worst case — six accumulated terms of limbs at the ends of
the normalised range — and pin the margin: |round-off| <= 2^-8 on random limbs of any magnitude, <= 0.2 on the coherent
patterns that maximise both the sums (6 * 4096 * 2^32) and the concentration of the spectrum."""
import numpy as np
import pytest

from _pkg import load_package

pytestmark = pytest.mark.gpu
N = 4096


def exact_negacyclic(a, g):
    """sum_r a_r * g_r mod X^N + 1 over the integers (int64 is enough: |sum| <= 8 * 4096 * 2^32 = 2^47)"""
    out = np.zeros(N, dtype=np.int64)
    for ar, gr in zip(a.astype(np.int64), g.astype(np.int64)):
        full = np.zeros(2 * N, dtype=np.int64)
        for i0 in range(0, N, 512):                      # blocked outer products: exact integer arithmetic
            blk = np.outer(ar[i0:i0 + 512], gr)
            for k in range(512):
                full[i0 + k:i0 + k + N] += blk[k]
        out += full[:N] - full[N:]
    return out


@pytest.fixture(scope="module")
def ram():
    pkg = load_package()
    return pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 4096)


def _patterns(rng, terms):
    lo, hi = -(1 << 16), (1 << 16) - 1
    yield "uniform", rng.integers(lo, hi + 1, (terms, N)), rng.integers(lo, hi + 1, (terms, N)), 2.0 ** -8
    yield "extremes, random signs", rng.choice([lo, hi], (terms, N)), rng.choice([lo, hi], (terms, N)), 2.0 ** -7
    yield "small", rng.integers(-3, 4, (terms, N)), rng.integers(lo, hi + 1, (terms, N)), 2.0 ** -12
    const = np.full((terms, N), lo)
    yield "every coefficient -2^16", const, const, 0.2
    coh = const.copy()
    coh[:, 0] = hi
    yield "coherent", coh, const, 0.2
    alt = const * np.where(np.arange(N) % 2 == 0, 1, -1)
    yield "alternating", alt, const, 0.05


@pytest.mark.parametrize("singles", [False, True])
@pytest.mark.parametrize("terms", [2, 6])
def test_products_round_to_the_exact_integers(ram, terms, singles):
    rng = np.random.default_rng(7 + terms)
    for name, a, g, bound in _patterns(rng, terms):
        raw = ram.selftest_convolve(a, g, singles=singles)
        want0 = exact_negacyclic(a, g)
        want1 = exact_negacyclic(a, g[np.arange(terms) ^ 1])
        assert np.abs(want0).max() < 2 ** 47
        err = max(np.abs(raw[0] - want0).max(), np.abs(raw[1] - want1).max())
        assert err <= bound, (name, terms, singles, err)
        assert np.array_equal(np.rint(raw[0]).astype(np.int64), want0), name
        assert np.array_equal(np.rint(raw[1]).astype(np.int64), want1), name


def test_pairs_and_singles_agree_bit_for_bit(ram):
    """a polynomial's transform does not depend on its partner: the same lanes run the same instructions on its points"""
    rng = np.random.default_rng(11)
    a = rng.integers(-(1 << 16), 1 << 16, (4, N))
    g = rng.integers(-(1 << 16), 1 << 16, (4, N))
    assert np.array_equal(ram.selftest_convolve(a, g, singles=False), ram.selftest_convolve(a, g, singles=True))
