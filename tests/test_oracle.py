"""CPU tests of the oracle (the checker itself): committed golden vectors, algebraic properties of
the limb arithmetic, and the reference's functional contract
(/root/reference/examples/fhe-ram.rs:97-176) at several sizes."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


def replay(po, params, inp):
    """Runs the example flow from recorded inputs; returns the outputs dict."""
    o = po.Oracle(params)
    evk = {k: np.ascontiguousarray(inp[k]) for k in ("gal_els", "atk_glwe", "atk_ggsw_inv", "tsk")}
    keys = o.keys_prepare(evk)
    addr = o.address_new(np.ascontiguousarray(inp["addr"]))
    ram = o.ram_new()
    ram.load(np.ascontiguousarray(inp["rows"]))
    out = {"read": ram.read(addr, keys), "rpw": ram.read_prepare_write(addr, keys), "rows_after_rpw": ram.store()}
    t = ram.tree(0)
    if t is not None:
        out["tree_after_rpw"] = t
    ram.write(np.ascontiguousarray(inp["w"]), addr, keys)
    out["rows_after_write"] = ram.store()
    out["readback"] = ram.read(addr, keys)
    return o, out


@pytest.mark.parametrize("n", [16, 64])
def test_golden_flow_small_n(po, n):
    z = np.load(os.path.join(GOLD, f"flow_n{n}.npz"))
    meta = z["meta"]
    params = po.OParams(log_n=int(meta[0]), max_addr=int(meta[1]), word_size=int(meta[2]), decomp_n=[int(x) for x in meta[4:]])
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    o, out = replay(po, params, inp)
    for k in z.files:
        if k.startswith("out_"):
            assert np.array_equal(out[k[4:]], z[k]), k
    # setup side is reproducible from the recorded seed as well
    seed = int(meta[3])
    assert np.array_equal(o.secret_gen(seed), inp["sk"])
    assert np.array_equal(o.evk_gen(inp["sk"], seed + 1, seed + 2)["atk_glwe"], inp["atk_glwe"])


@pytest.mark.parametrize("key", ["4096", "16384", "readme_16384"])
def test_golden_digests_n4096(po, key):
    """readme_*: the parameter block of README.md:17-27 (K_PT = 9, 5-limb trace keys)"""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    d = json.load(open(os.path.join(GOLD, "digests_n4096.json")))[key]
    max_addr = d.get("max_addr") or int(key)
    inp, out, o = make_golden.flow(po.OParams(max_addr=max_addr, word_size=d["word_size"], **d.get("params", {})), d["seed"])
    assert {k: sha(v) for k, v in inp.items()} == d["inputs"]
    assert {k: sha(v) for k, v in out.items()} == d["outputs"]
    assert o.max_big() < 1 << 47                      # exact below the HIP prime / 2


def test_znx_known_answers(po):
    kat = json.load(open(os.path.join(GOLD, "znx_kat.json")))
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    for c in kat["cases"]:
        a = np.array(c["in"], dtype=np.int64)
        if c["op"] == "big_normalize":
            got = o.big_normalize(a, c["res_size"])
        elif c["op"] == "glwe_rsh":
            got = o.glwe_rsh(c["k"], a)
        elif c["op"] == "glwe_rotate":
            got = o.glwe_rotate(c["k"], a)
        else:
            got = o.poly_automorphism(c["g"], a)
        assert np.array_equal(got, np.array(c["out"], dtype=np.int64)), c["op"]


def value_of(limbs, base2k=17):
    """Integer sum_j x_j 2^(base2k (size-1-j)) per coefficient (python ints)."""
    size = limbs.shape[0]
    return [sum(int(limbs[j, i]) << (base2k * (size - 1 - j)) for j in range(size)) for i in range(limbs.shape[1])]


def test_normalize_preserves_value_and_range(po):
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    rng = np.random.default_rng(1)
    a = rng.integers(-(1 << 47), 1 << 47, size=(4, 16), dtype=np.int64)
    r = o.big_normalize(a, 4)
    assert r.min() >= -(1 << 16) and r.max() < (1 << 16)
    mod = 1 << (17 * 4)
    assert [x % mod for x in value_of(r)] == [x % mod for x in value_of(a)]
    # dropping the last limb rounds to nearest (carry only)
    r3 = o.big_normalize(a, 3)
    for v4, v3 in zip(value_of(r), value_of(r3)):
        assert ((v4 - (v3 << 17)) + (1 << 16)) % (1 << (17 * 4)) < (1 << 17) or abs((v4 - (v3 << 17)) % mod) <= (1 << 16)
    # normalising a normalised vector is the identity
    assert np.array_equal(o.big_normalize(r, 4), r)


def test_rsh1_is_exact_halving_with_round_half_up(po):
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    rng = np.random.default_rng(2)
    g = rng.integers(-(1 << 17), 1 << 17, size=3 * 2 * 16, dtype=np.int64)
    out = o.glwe_rsh(1, g)
    assert out.min() >= -(1 << 16) and out.max() < (1 << 16)          # output is normalised
    mod = 1 << 51
    gi, oi = g.reshape(3, 2, 16), out.reshape(3, 2, 16)
    for col in range(2):
        for x, y in zip(value_of(gi[:, col]), value_of(oi[:, col])):
            assert (y - (-(-x // 2))) % mod == 0                       # ceil(x/2) mod 2^51


def test_rotate_and_automorphism_algebra(po):
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    rng = np.random.default_rng(3)
    g = rng.integers(-(1 << 16), 1 << 16, size=3 * 2 * 16, dtype=np.int64)
    assert np.array_equal(o.glwe_rotate(-5, o.glwe_rotate(5, g)), g)
    assert np.array_equal(o.glwe_rotate(16, g), -g)                    # X^N = -1
    assert np.array_equal(o.glwe_rotate(32, g), g)
    a = rng.integers(-9, 9, size=16, dtype=np.int64)
    b = rng.integers(-9, 9, size=16, dtype=np.int64)
    for gal in (-1, 5, 25):
        lhs = o.poly_automorphism(gal, po.Oracle.negacyclic_schoolbook(a, b))
        rhs = po.Oracle.negacyclic_schoolbook(o.poly_automorphism(gal, a), o.poly_automorphism(gal, b))
        assert np.array_equal(lhs, rhs)                                # phi is a ring automorphism
    assert np.array_equal(o.poly_automorphism(-1, o.poly_automorphism(-1, a)), a)


@pytest.mark.parametrize("log_n", [4, 6, 8, 12])
def test_ntt_matches_schoolbook(po, log_n):
    o = po.Oracle(po.OParams(log_n=log_n, max_addr=1 << log_n, decomp_n=[log_n // 2, log_n - log_n // 2]))
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    a = rng.integers(-(1 << 16), 1 << 16, size=n, dtype=np.int64)
    b = rng.integers(-(1 << 16), 1 << 16, size=n, dtype=np.int64)
    if log_n == 12:
        a[:] = -(1 << 16)
        b[:] = -(1 << 16)                                              # worst case magnitude N * 2^32
    assert np.array_equal(o.negacyclic_ntt(a, b), po.Oracle.negacyclic_schoolbook(a, b))


def test_exact_product_is_what_a_float64_fft_rounds_to(po):
    """The reference multiplies limb polynomials with an f64 FFT backend and rounds to i64 (examples/fhe-ram.rs:3-7,
    SURVEY.md A.9).  Emulated with numpy's complex128 FFT (negacyclic through the psi-twist, full-size
    transform), the rounded result equals the oracle's exact product for random
    limbs, for the all-extreme-digit worst case, and for a 6-term accumulation as in an external product —
    i.e. the exact-integer arithmetic restated here is the arithmetic the reference performs whenever its FFT is
    accurate, which the magnitude bound 6*2^44 << 2^53 guarantees."""
    n = 4096
    o = po.Oracle(po.OParams(max_addr=1 << 12))
    rng = np.random.default_rng(9)
    twist = np.exp(1j * np.pi * np.arange(n) / n)

    def fft_negacyclic(a, b):
        fa, fb = np.fft.fft(a * twist), np.fft.fft(b * twist)
        return np.fft.ifft(fa * fb) / twist

    pairs = [(rng.integers(-(1 << 16), 1 << 16, size=n, dtype=np.int64), rng.integers(-(1 << 16), 1 << 16, size=n, dtype=np.int64))
             for _ in range(6)]
    worst = (np.full(n, -(1 << 16), dtype=np.int64), np.full(n, -(1 << 16), dtype=np.int64))
    for a, b in pairs[:2] + [worst]:
        c = fft_negacyclic(a.astype(np.float64), b.astype(np.float64))
        assert np.max(np.abs(c.imag)) < 0.25 and np.max(np.abs(c.real - np.rint(c.real))) < 0.25
        assert np.array_equal(np.rint(c.real).astype(np.int64), o.negacyclic_ntt(a, b))
    acc_f = sum(np.fft.fft(a * twist) * np.fft.fft(b * twist) for a, b in pairs)       # accumulate in the transform domain
    acc = np.fft.ifft(acc_f) / twist
    want = sum(o.negacyclic_ntt(a, b) for a, b in pairs)
    assert np.max(np.abs(acc.real - np.rint(acc.real))) < 0.25
    assert np.array_equal(np.rint(acc.real).astype(np.int64), want)


def test_galois_elements(po):
    assert [int(po.lib().fo_galois_element(12, i)) for i in range(12)] == \
        [-1, 5, 25, 625, 5601, 4033, 3969, 7937, 7681, 7169, 6145, 4097]
    from _pkg import load_package
    assert load_package().galois_elements(12) == [-1, 5, 25, 625, 5601, 4033, 3969, 7937, 7681, 7169, 6145, 4097]


def test_cast_u8_to_signed(po):  # examples/fhe-ram.rs:25-32
    f = po.Oracle.cast_u8_to_signed
    assert f(0xFF, 8) == -1 and f(0x7F, 8) == 127 and f(0x80, 8) == -128
    assert f(7, 3) == -1 and f(3, 3) == 3 and f(4, 3) == -4 and f(0xF8 | 2, 3) == 2


def test_op_counts_match_survey_appendix_b(po):
    """EP / KS counts per op follow from the reference's control flow."""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    p = po.OParams(max_addr=1 << 14, word_size=4)
    o = po.Oracle(p)
    sk = o.secret_gen(1)
    keys = o.keys_prepare(o.evk_gen(sk, 2, 3))
    rows = o.ram_encrypt(np.zeros(p.max_addr * 4, dtype=np.uint8), sk, 4, 5)
    addr = o.address_new(o.address_encrypt(12345, sk, 6, 7))
    ram = o.ram_new()
    ram.load(rows)
    o.reset_stats()
    ram.read(addr, keys)
    assert (o.counters()["ep"], o.counters()["ks"]) == (68, 220)
    o.reset_stats()
    ram.read_prepare_write(addr, keys)
    assert (o.counters()["ep"], o.counters()["ks"]) == (68, 220)
    o.reset_stats()
    ram.write(rows[:, 0], addr, keys)
    assert (o.counters()["ep"], o.counters()["ks"]) == (68, 432 + 30)


def test_state_machine_and_size_asserts(po):  # ram.rs:393-396,555-558,243; coordinate_prepared.rs:134
    p = po.OParams(log_n=4, max_addr=1 << 6, decomp_n=[2, 2], word_size=2)
    o = po.Oracle(p)
    sk = o.secret_gen(1)
    evk = o.evk_gen(sk, 2, 3)
    keys = o.keys_prepare(evk)
    rows = o.ram_encrypt(np.arange(128, dtype=np.uint8), sk, 4, 5)
    addr = o.address_new(o.address_encrypt(9, sk, 6, 7))
    ram = o.ram_new()
    with pytest.raises(po.OracleError, match="unitialized memory"):
        ram.read(addr, keys)
    ram.load(rows)
    with pytest.raises(po.OracleError, match="requires calling Memory.read_prepare_write"):
        ram.write(rows[:, 0], addr, keys)
    ram.read_prepare_write(addr, keys)
    with pytest.raises(po.OracleError, match="requires calling Memory.write"):
        ram.read(addr, keys)
    with pytest.raises(po.OracleError):
        ram.write(rows[:1, 0], addr, keys)
    with pytest.raises(po.OracleError, match="max_addr"):
        o.ram_encrypt(np.arange(64, dtype=np.uint8), sk, 4, 5)


def test_all_core_variant_equals_one_thread(po):
    """The OpenMP variant of the oracle's Ram ops (cpu_baseline_allcores in bench.py, golden digests of the
    full sizes) runs sub-RAMs and rows concurrently; every ciphertext sees the same operations, so the
    results must be identical to the sequential restatement."""
    import numpy as np
    for max_addr in (5 * 4096, 2 * 4096, 7 * 4096 - 100):
        _all_core_case(po, max_addr)


def _all_core_case(po, max_addr):
    import numpy as np
    outs = []
    for th in (1, 4):
        o = po.Oracle(po.OParams(max_addr=max_addr, word_size=2)).set_threads(th)
        sk = o.secret_gen(1)
        keys = o.keys_prepare(o.evk_gen(sk, 2, 3))
        rng = np.random.default_rng(4)
        data = rng.integers(0, 256, size=max_addr * 2, dtype=np.uint8)
        ram = o.ram_new()
        ram.load(o.ram_encrypt(data, sk, 5, 6))
        addr = o.address_new(o.address_encrypt(max_addr - 4096 + 17, sk, 7, 8))
        w = np.stack([o.glwe_encrypt_coeff0(5 + i, sk, 9 + i, 19 + i) for i in range(2)])
        r = [ram.read(addr, keys), ram.read_prepare_write(addr, keys), ram.store(), ram.tree(0)]
        ram.write(w, addr, keys)
        r += [ram.store(), ram.tree(0), ram.read(addr, keys)]
        outs.append((r, o.counters(), o.max_big()))
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)


def test_committed_digests_cover_the_full_sizes():
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests_n4096.json")))
    for size in ("4096", "16384", "262144", "2097152", "readme_16384", "readme_262144"):      # BASELINE.json configs[0], source default, configs[2..3], configs[4]; README.md:17-34 block
        assert {"read", "rpw", "rows_after_rpw", "rows_after_write", "readback"} <= set(d[size]["outputs"])
        assert d[size]["max_big_log2"] < 47


def test_rsh1_closed_form(po):
    """vec_znx_rsh(1) (oracle/znx.hpp rsh_inplace; GLWE::trace / GLWEPacker, ram.rs:435,457) equals the centred base-2^17
    digits of ceil(A / 2), A = a_0*2^34 + a_1*2^17 + a_2 — the form the HIP trace chains hand their intermediate
    ciphertexts over in (csrc/kernels.hpp ks_trace_y).  Every floating-point operation of that form is exact (powers of
    two, +0.5, floor, differences of exact values), so numpy float64 reproduces the device arithmetic; compared digit for
    digit with the oracle on random limbs and on the extremes (+-2^16, the un-normalised +2^16 a rotation's negation
    produces, odd / even tails)."""
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    n, B = 16, 2.0 ** 17
    rng = np.random.default_rng(1)

    def closed_form(g):
        x = g.reshape(3, 2, n).astype(np.float64)
        A = (x[0] * B + x[1]) * B + x[2]
        assert np.all(np.abs(A) < 2.0 ** 51)
        Y = np.floor(A * 0.5 + 0.5)
        q1 = np.floor(Y / B + 0.5)
        q2 = np.floor(q1 / B + 0.5)
        assert np.all(np.abs(q2) < 2 ** 16)                 # the second quotient IS the top digit: no wrap (ks_trace_y relies on it)
        return np.stack([q2, q1 - q2 * B, Y - q1 * B]).astype(np.int64).reshape(-1)

    edge = np.array([-(1 << 16), (1 << 16) - 1, 1 << 16, 0, 1, -1, 2, -2, 65535, -65535])
    for it in range(1500):
        mode = it % 5
        if mode == 0:
            g = rng.integers(-(1 << 16), 1 << 16, size=3 * 2 * n, dtype=np.int64)
        elif mode == 1:
            g = rng.choice(edge, size=3 * 2 * n).astype(np.int64)
        elif mode == 2:
            g = rng.integers(-(1 << 16), (1 << 16) + 1, size=3 * 2 * n, dtype=np.int64)
            g[:2 * n] = rng.choice(np.array([1 << 16, -(1 << 16)]), size=2 * n)
        elif mode == 3:
            g = rng.integers(-3, 4, size=3 * 2 * n, dtype=np.int64)
        else:
            g = rng.integers(-(1 << 16), (1 << 16) + 1, size=3 * 2 * n, dtype=np.int64)
            g[4 * n:] = rng.choice(edge, size=2 * n)
        assert np.array_equal(closed_form(g), o.glwe_rsh(1, g)), (it, mode)


# ---- round 4: the closed-form normalisation of the HIP chain kernels (csrc/kernels.hpp: fold_limb, window51, ks_trace_z / ks_trace_l,
# ep_step_r, k_pair_z), restated in numpy float64 — every operation of it is exact on the device, so numpy reproduces the device
# arithmetic — against the oracle's limb-by-limb code
_B = 2.0 ** 17
_B2 = 2.0 ** 34
_M = 2.0 ** 51
_C = 2.0 ** 16 + 2.0 ** 33 + 2.0 ** 50


def _carry(x):
    return np.floor(x / _B + 0.5)


def _cmod(x, m):
    return x - m * np.floor(x / m + 0.5)


def _window51(v):
    return v - _M * np.floor((v + _C) / _M)


def _digits(a):
    """balanced base-2^17 digits (d0, d1, d2) of an integer in the window, as the kernels' take_digit chain produces them"""
    q1 = _carry(a)
    d2 = a - q1 * _B
    q2 = _carry(q1)
    d1 = q1 - q2 * _B
    return np.stack([q2, d1, d2])


def test_closed_form_normalisation_of_big_limbs(po):
    """vec_znx_big_normalize of SK un-normalised limbs into 3 digits (4 -> 3 and 5 -> 3: the trace keys of both parameter blocks; the
    external product's 4 -> 3) equals the digits of window51(e + big_2 + cmod34(big_1) 2^17 + cmod17(big_0) 2^34) with e the rounded carry
    of the extra limbs: random limbs up to the worst-case magnitude 2^47, values on rounding ties, all-extreme limbs."""
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    n = 16
    rng = np.random.default_rng(4)
    for sk in (4, 5):
        for it in range(600):
            mode = it % 4
            if mode == 0:
                big = rng.integers(-(1 << 47) + 1, 1 << 47, size=(sk, n), dtype=np.int64)
            elif mode == 1:      # rounding ties of the carries: multiples of 2^16
                big = rng.integers(-(1 << 30), 1 << 30, size=(sk, n), dtype=np.int64) * (1 << 16)
            elif mode == 2:
                big = rng.choice(np.array([(1 << 47) - 1, -(1 << 47) + 1, 0, 1 << 16, -(1 << 16), (1 << 16) - 1, 1 << 33, -(1 << 33)]), size=(sk, n)).astype(np.int64)
            else:
                big = rng.integers(-(1 << 20), 1 << 20, size=(sk, n), dtype=np.int64)
            want = o.big_normalize(big, 3)
            b = big.astype(np.float64)
            e = np.zeros(n)
            for j in range(sk - 1, 2, -1):          # the extra limbs, least significant first
                e = _carry(b[j] + e)
            v = e + b[2] + _cmod(b[1], _B2) * _B + _cmod(b[0], _B) * _B2
            assert np.all(np.abs(v) < 2.0 ** 53)
            got = _digits(_window51(v)).astype(np.int64)
            assert np.array_equal(got, want), (sk, it, mode)


def test_closed_form_trace_step_post_step(po):
    """The post-step of a trace step, a <- rsh1(a) + phi(KS(rsh1(a))) normalised (ram.rs:457,514,616): limbs of rsh1(a) added to the big
    limbs, normalised limb by limb by the oracle, against window51(Y + e + big_2 + ...) with Y = ceil(A / 2) (ks_trace_z)."""
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    n = 16
    rng = np.random.default_rng(5)
    for it in range(800):
        g = rng.integers(-(1 << 16), 1 << 16, size=(3, n), dtype=np.int64)
        if it % 3 == 1:
            g = rng.choice(np.array([-(1 << 16), (1 << 16) - 1, 1 << 16, 0, 1, -1]), size=(3, n)).astype(np.int64)
        # rsh1 of one column through the oracle (a GLWE has two columns: use the same limbs in both)
        glwe = np.stack([g, g], axis=1).reshape(-1)
        x = o.glwe_rsh(1, glwe).reshape(3, 2, n)[:, 0, :]
        big = rng.integers(-(1 << 47) + 1, 1 << 47, size=(4, n), dtype=np.int64)
        summed = big.copy()
        summed[:3] += x
        want = o.big_normalize(summed, 3)
        a = (g[0].astype(np.float64) * _B + g[1]) * _B + g[2]
        y = np.floor(a * 0.5 + 0.5)
        b = big.astype(np.float64)
        v = y + _carry(b[3]) + b[2] + _cmod(b[1], _B2) * _B + _cmod(b[0], _B) * _B2
        got = _digits(_window51(v)).astype(np.int64)
        assert np.array_equal(got, want), it


def test_closed_form_pair_combine_and_write_elementwise(po):
    """k_pair_z: rsh1 of a limb-wise SUM / DIFFERENCE of two ciphertexts (un-normalised limbs up to +-2^17) equals the digits of
    window51(ceil((A(a) +- A(b)) / 2)), and normalize(u - tmp) of two normalised limb vectors equals the digits of window51(U - T);
    k_write_chain: normalize(h - t + b) equals the digits of window51(A(h) - A(t) + A(b)) (ram.rs:617,625-626)."""
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    n = 16
    rng = np.random.default_rng(6)

    def whole(l):
        l = l.astype(np.float64)
        return (l[0] * _B + l[1]) * _B + l[2]
    edge = np.array([-(1 << 16), (1 << 16) - 1, 1 << 16, 0, 1, -1])
    for it in range(800):
        pick = (lambda: rng.integers(-(1 << 16), 1 << 16, size=(3, n), dtype=np.int64)) if it % 3 else (lambda: rng.choice(edge, size=(3, n)).astype(np.int64))
        a, b, c = pick(), pick(), pick()
        for sgn in (1, -1):
            v = a + sgn * b
            want = o.glwe_rsh(1, np.stack([v, v], axis=1).reshape(-1)).reshape(3, 2, n)[:, 0, :]
            got = _digits(_window51(np.floor((whole(a) + sgn * whole(b)) * 0.5 + 0.5))).astype(np.int64)
            assert np.array_equal(got, want), (it, sgn)
        want = o.glwe_normalize(np.stack([a - b, a - b], axis=1).reshape(-1)).reshape(3, 2, n)[:, 0, :]
        assert np.array_equal(_digits(_window51(whole(a) - whole(b))).astype(np.int64), want), it
        want = o.glwe_normalize(np.stack([a - b + c, a - b + c], axis=1).reshape(-1)).reshape(3, 2, n)[:, 0, :]
        assert np.array_equal(_digits(_window51(whole(a) - whole(b) + whole(c))).astype(np.int64), want), it
