"""Directed device tests of the FP64 modular arithmetic (csrc/ntt_dev.hpp): mulmod / macmod / reduce
and the lazy-reduction bounds of the 4096-point transforms, driven with adversarial operands and compared with exact
integer arithmetic (Python ints) on the host.  The end-to-end parity tests only ever see operands of typical
magnitude; these drive the documented worst cases: |a| up to 32p, |b| <= p/2, products at 2^100..2^101, quotients
a*b/p on rounding ties, all-aligned inputs for the 12-stage drift and for the first inverse pass without the initial
reduction."""
import ctypes as C

import numpy as np
import pytest

from _pkg import load_package

N = 4096


def _consts(pkg):
    p, psi = C.c_uint64(), C.c_uint64()
    pkg.library().fheram_selftest_constants(C.byref(p), C.byref(psi))
    return int(p.value), int(psi.value)


def test_constants_are_a_prime_and_a_primitive_root():
    """runs without a GPU: p prime (deterministic Miller-Rabin), p = 1 mod 2N, psi of order exactly 2N"""
    pkg = load_package()
    p, psi = _consts(pkg)
    assert p == (1 << 48) + 57345 and p % (2 * N) == 1
    d, s = p - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, p)
        if x in (1, p - 1):
            continue
        for _ in range(s - 1):
            x = x * x % p
            if x == p - 1:
                break
        else:
            raise AssertionError("p is composite")
    assert pow(psi, 2 * N, p) == 1 and pow(psi, N, p) == p - 1


def _brv(x, bits=12):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def _tables(p, psi):
    W = [pow(psi, _brv(i), p) for i in range(N)]
    Winv = [pow(w, p - 2, p) for w in W]
    return W, Winv


def ref_fwd(x, p, W):
    """standard in-place negacyclic Cooley-Tukey transform, psi^bitrev twiddles; output in bit-reversed order"""
    x = [int(v) % p for v in x]
    t, m = N, 1
    while m < N:
        t //= 2
        for i in range(m):
            w = W[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = x[j], x[j + t] * w % p
                x[j], x[j + t] = (u + v) % p, (u - v) % p
        m *= 2
    return x


def ref_inv_xN(x, p, Winv):
    """Gentleman-Sande inverse of ref_fwd WITHOUT the 1/N factor"""
    x = [int(v) % p for v in x]
    t, m = 1, N // 2
    while m >= 1:
        for i in range(m):
            w = Winv[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = x[j], x[j + t]
                x[j], x[j + t] = (u + v) % p, (u - v) * w % p
        t *= 2
        m //= 2
    return x


def centred(v, p):
    v %= p
    return v - p if v > p // 2 else v


@pytest.fixture(scope="module")
def ram():
    pkg = load_package()
    return pkg.Ram.new_from_ram_params(1, [3, 3, 3, 3], 1 << 12)


def _as_int(d):
    """a double that must hold an exact integer"""
    assert np.all(np.isfinite(d)) and np.all(d == np.rint(d))
    return [int(v) for v in d]


@pytest.mark.gpu
def test_mulmod_macmod_reduce_on_adversarial_operands(ram):
    pkg = load_package()
    p, _ = _consts(pkg)
    rng = np.random.default_rng(2026)
    A, B, ACC = [], [], []

    def add(a, b, acc=0):
        assert abs(a) < 1 << 53 and abs(b) <= p // 2 + 1 and abs(acc) < 1 << 52
        A.append(a)
        B.append(b)
        ACC.append(acc)

    half = p // 2
    # (i) random operands over the whole documented range: |a| < 32p = 2^53, |b| <= p/2
    for _ in range(20000):
        add(int(rng.integers(-(1 << 53) + 1, 1 << 53)), int(rng.integers(-half, half + 1)), int(rng.integers(-(1 << 50), 1 << 50)))
    # (ii) extremes: largest magnitudes, products at 2^100 .. 2^101
    for a in ((1 << 53) - 1, -(1 << 53) + 1, (1 << 53) - 2, 16 * p, -16 * p, 11 * p + 12345, 27 * p + 1, 31 * p + p // 2):
        for b in (half, -half, half - 1, 1 - half, 1, -1, 0, 57345, (1 << 47) - 1, -(1 << 47)):
            add(a, b, 4 * p - 7)
    # (iii) quotient on a rounding tie: a*b = t (mod p) with t next to +-p/2, and next to 0 (q exact, remainder tiny)
    for _ in range(4000):
        b = int(rng.integers(-half, half + 1)) or 1
        binv = pow(b % p, p - 2, p)
        t = int(rng.choice([half, half + 1, half - 1, half + 2, -half, -half - 1, 0, 1, -1, 2]))
        a0 = (t % p) * binv % p
        k = int(rng.integers(-31, 31))
        a = a0 + k * p
        if abs(a) >= 1 << 53:
            a = a0 - p
        add(a, b, int(rng.integers(-3 * p, 3 * p)))
    a, b, acc = (np.array(v, dtype=np.float64) for v in (A, B, ACC))
    assert all(int(x) == y for x, y in zip(a, A)) and all(int(x) == y for x, y in zip(b, B))   # exactly representable
    mul, mac, red = ram.selftest_modarith(a, b, acc)
    mul_i, mac_i, red_i = _as_int(mul), _as_int(mac), _as_int(red)
    worst = 0.0
    for i, (ai, bi, ci) in enumerate(zip(A, B, ACC)):
        assert (mul_i[i] - ai * bi) % p == 0, ("mulmod residue", ai, bi, mul_i[i])
        # |result| <= (0.5 + 3 * 2^-53 * |a*b| / p) p + the low part folded in exactly
        bound = (0.5 + 3.0 * 2.0 ** -53 * abs(ai * bi) / p) * p + 2
        assert abs(mul_i[i]) <= bound, ("mulmod magnitude", ai, bi, mul_i[i], bound)
        worst = max(worst, abs(mul_i[i]) / p)
        assert (mac_i[i] - ci - ai * bi) % p == 0, ("macmod residue", ai, bi, ci, mac_i[i])
        assert abs(mac_i[i] - ci) <= bound, ("macmod magnitude", ai, bi, ci, mac_i[i])
        assert (red_i[i] - ai) % p == 0 and abs(red_i[i]) <= p // 2 + 64, ("reduce", ai, red_i[i])
    print(f"mulmod: {len(A)} operand pairs exact; largest |result| = {worst:.3f} p")
    assert worst < 2.1        # operands of 32p x p/2: (0.5 + 1.5) p


@pytest.mark.gpu
def test_forward_transform_residues_and_drift_bound(ram):
    pkg = load_package()
    p, psi = _consts(pkg)
    W, _ = _tables(p, psi)
    rng = np.random.default_rng(7)
    lim = 1 << 17          # forward inputs of the kernels: limbs (|x| <= 2^16) and pre-stepped sums, < 2^18
    polys = [np.full(N, lim), np.full(N, -lim), np.where(np.arange(N) % 2 == 0, lim, -lim),
             np.where(rng.integers(0, 2, N) == 0, lim, -lim), rng.integers(-lim, lim + 1, N), rng.integers(-lim, lim + 1, N),
             np.eye(1, N, 0)[0] * lim, np.eye(1, N, N - 1)[0] * -lim, (1 << 18) - 1 - np.zeros(N)]
    x = np.array(polys, dtype=np.float64)
    out = ram.selftest_ntt(0, x)
    worst = 0.0
    for k in range(x.shape[0]):
        got = _as_int(out[k])
        want = ref_fwd([int(v) for v in x[k]], p, W)
        assert all((g - w) % p == 0 for g, w in zip(got, want)), f"forward transform residues differ (poly {k})"
        worst = max(worst, max(abs(g) for g in got) / p)
    print(f"forward transform: largest unreduced output = {worst:.2f} p (documented bound 11 p)")
    assert worst < 11.0


@pytest.mark.gpu
@pytest.mark.parametrize("direction,amp", [(1, 15.9), (2, 3.05)])
def test_inverse_transform_exact_at_the_documented_input_bound(ram, direction, amp):
    """dir 1: inputs up to 16p (what ntt_inv documents) with the initial reduction; dir 2: the key-switch form without
    it, inputs up to 3.06p (three MAC terms of <= 1.02p).  All-aligned signs drive slot 0 of the first pass to E * I."""
    pkg = load_package()
    p, psi = _consts(pkg)
    _, Winv = _tables(p, psi)
    rng = np.random.default_rng(11 + direction)
    top = int(amp * p)
    polys = []
    r = rng.integers(0, p, N, dtype=np.int64)
    for mode in ("all+", "all-", "alt", "rand", "rand", "edge"):
        if mode == "all+":       # random residues, every value in (top - p, top]
            v = [int(ri) + (top - int(ri)) // p * p for ri in r]
        elif mode == "all-":
            v = [-(int(ri) + (top - int(ri)) // p * p) for ri in r]
        elif mode == "alt":
            v = [(top - int(ri) % (p // 4)) * (1 if i % 2 == 0 else -1) for i, ri in enumerate(r)]
        elif mode == "rand":
            v = [int(x_) for x_ in rng.integers(-top, top + 1, N)]
        else:                    # exactly at the bound, and on multiples of p
            v = [top if i % 3 == 0 else (-top if i % 3 == 1 else (i % 7 - 3) * p) for i in range(N)]
        assert max(abs(vi) for vi in v) <= top
        polys.append(v)
    x = np.array(polys, dtype=np.float64)
    assert all(int(a) == b for a, b in zip(x.ravel(), [vi for v in polys for vi in v]))
    out = ram.selftest_ntt(direction, x)
    for k, v in enumerate(polys):
        got = _as_int(out[k])
        want = ref_inv_xN(v, p, Winv)        # natural order: position i of the in-place result
        assert all((g - w) % p == 0 for g, w in zip(got, want)), f"inverse transform residues differ (poly {k})"
        assert max(abs(g) for g in got) <= p // 2 + 64, "inverse transform output is not centred"


@pytest.mark.gpu
def test_transform_round_trip_is_N_times_identity(ram):
    """independent of the host-side reference: inv(fwd(x)) = N x (mod p) on worst-case limb inputs"""
    pkg = load_package()
    p, _ = _consts(pkg)
    rng = np.random.default_rng(3)
    x = np.array([rng.integers(-(1 << 17), (1 << 17) + 1, N), np.full(N, 1 << 17), np.full(N, -(1 << 17))], dtype=np.float64)
    back = ram.selftest_ntt(1, ram.selftest_ntt(0, x))
    for k in range(x.shape[0]):
        got = _as_int(back[k])
        assert all((g - N * int(v)) % p == 0 for g, v in zip(got, x[k]))
