"""SURVEY.md 8(f) N4 — Address::set_from_fheuint (/root/reference/src/conversion.rs:18-82).

The reference delegates to poulpy-schemes' scalar_to_ggsw_blind_rotation (un-vendored); oracle/ restates its CONTRACT
with CMux chains (oracle/fheram_oracle.hpp: address_set_from_fheuint).  These tests are the reference's own contract test
(conversion.rs:100-220) at the RAM's parameters: every derived digit decrypts, row by row and column by column, to
X^{((k >> bit_rsh) mod 2^bit_mask) << bit_lsh} on the gadget with noise below max_noise (:184-192, :215), and an
address derived with the RAM's sign convention reads the right word.  CPU: the oracle.  GPU: the HIP path, bit-exact
against the oracle and through Ram.read."""
import numpy as np
import pytest

N, B2K, K_ADDR, SIZE, DNUM = 4096, 17, 68, 4, 3


def max_noise(col_i):
    """conversion.rs:184-192 with size = 4, base2k = 17, SIGMA = 3.2, log_n = 12"""
    noise = -(SIZE * B2K) + np.log2(3.2) + 3.0 + 0.5 * 12
    return noise + (0.5 * 12 if col_i != 0 else 0.0)


def check_digits(o, digits, sk, k, base2d, sign):
    """ggsw.assert_noise(module, sk, pt, max_noise) for every digit (conversion.rs:200-218)"""
    s = np.array(sk, dtype=object)
    bit_rsh, d = 0, 0
    worst = -1e9
    for base1d in base2d:
        bit_lsh = 0
        for bit_mask in base1d:
            rot = (((k >> bit_rsh) & ((1 << bit_mask) - 1)) << bit_lsh)
            if not sign:
                rot = -rot
            pt = np.zeros(N, dtype=object)
            r_ = rot % (2 * N)
            pt[r_ % N] = -1 if r_ >= N else 1                      # X^rot in Z[X]/(X^N + 1)
            pt_s = np.zeros(N, dtype=object)                       # pt * s
            for i in np.nonzero(pt)[0]:
                for j in np.nonzero(s)[0]:
                    q = int(i + j)
                    pt_s[q % N] += (-1 if q >= N else 1) * pt[i] * s[j]
            g = digits[d].reshape(DNUM, 2, SIZE * 2 * N)
            for r in range(DNUM):
                for ci in range(2):
                    ph = o.glwe_phase(g[r, ci], sk)         # normalised limbs [4][N] of body + mask*s
                    val = np.zeros(N, dtype=object)
                    for j in range(SIZE):
                        val = val * (1 << B2K) + np.array([int(x) for x in ph[j]], dtype=object)
                    want = (pt if ci == 0 else pt_s) * (1 << (SIZE * B2K - (r + 1) * B2K))
                    diff = val - want
                    half, mod = 1 << (SIZE * B2K - 1), 1 << (SIZE * B2K)
                    diff = np.array([((int(x) + half) % mod) - half for x in diff], dtype=object)
                    m = max(abs(int(x)) for x in diff)
                    noise = (np.log2(m) if m else -np.inf) - SIZE * B2K
                    worst = max(worst, noise - max_noise(ci))
                    assert noise <= max_noise(ci), (d, r, ci, noise, max_noise(ci))
            bit_lsh += bit_mask
            bit_rsh += bit_mask
            d += 1
    return worst


@pytest.mark.parametrize("sign", [True, False])
def test_oracle_set_from_fheuint_contract(po, sign):
    o = po.Oracle(po.OParams(max_addr=1 << 14, word_size=1))
    sk = o.secret_gen(21)
    k = 0b10_110_011_101_010
    bits = o.fheuint_encrypt(k, 14, sk, 22, 23)
    digits = o.address_from_fheuint(bits, sign=sign)
    check_digits(o, digits, sk, k, [[3, 3, 3, 3], [2]], sign)
    assert o.max_big() < 1 << 47


def test_oracle_derived_address_reads_the_word(po):
    o = po.Oracle(po.OParams(max_addr=1 << 14, word_size=2))
    sk = o.secret_gen(31)
    keys = o.keys_prepare(o.evk_gen(sk, 32, 33))
    rng = np.random.default_rng(34)
    data = rng.integers(0, 256, size=(1 << 14) * 2, dtype=np.uint8)
    ram = o.ram_new()
    ram.load(o.ram_encrypt(data, sk, 35, 36))
    for k in (0, 1, 4095, 4096, 12345, (1 << 14) - 1):
        digits = o.address_from_fheuint(o.fheuint_encrypt(k, 14, sk, 37, 38), sign=False)   # Address::encrypt_sk's convention
        got = ram.read(o.address_new(digits), keys)
        for i in range(2):
            want = o.cast_u8_to_signed(int(data[i + 2 * k]), 3)
            v, nz = o.glwe_decrypt(got[i], want, sk)
            assert v == want and nz < -4, (k, i, v, want, nz)


@pytest.mark.gpu
def test_hip_set_from_fheuint_bit_exact_and_functional(po):
    from _pkg import load_package
    pkg = load_package()
    max_addr, ws = 1 << 18, 2
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
    sk = o.secret_gen(41)
    evk = o.evk_gen(sk, 42, 43)
    rng = np.random.default_rng(44)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    rows = o.ram_encrypt(data, sk, 45, 46)
    ram = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], max_addr)
    ram.load_encrypted(rows)
    keys = pkg.EvaluationKeysPrepared.from_dict(evk)
    dsk = pkg.GLWESecret(ram, sk)
    k = 0b101_110_011_101_010_001
    # host-encrypted integer: the device derives exactly the digits the oracle derives
    bits = o.fheuint_encrypt(k, 18, sk, 47, 48)
    fu = pkg.FheUintPrepared.from_host(ram, bits)
    for sign in (True, False):
        addr = pkg.Address.set_from_fheuint(ram, fu, sign=sign)
        want = o.address_from_fheuint(bits, sign=sign)
        got = np.stack(addr.digits)
        assert np.array_equal(got, want), f"sign={sign}: {np.count_nonzero(got != want)} limbs differ"
    check_digits(o, np.stack(pkg.Address.set_from_fheuint(ram, fu, sign=True).digits), sk, k, [[3, 3, 3, 3], [3, 3]], True)
    # integer encrypted on the device with the oracle's draws replayed: same ciphertexts, then the derived address reads the word
    fu2 = pkg.FheUintPrepared.encrypt_sk(ram, k, dsk, o.source(47), o.source(48), n_bits=18)
    assert np.array_equal(fu2.download(), bits)
    addr = pkg.Address.set_from_fheuint(ram, fu2, sign=False)
    out = ram.read(addr, keys)
    oram = o.ram_new()
    oram.load(rows)
    assert np.array_equal(out, oram.read(o.address_new(o.address_from_fheuint(bits, sign=False)), o.keys_prepare(evk)))
    for i in range(ws):
        w_ = o.cast_u8_to_signed(int(data[i + ws * k]), 3)
        v, nz = o.glwe_decrypt(out[i], w_, sk)
        assert v == w_ and nz < -4, (v, w_, nz)
    # misuse
    small = pkg.FheUintPrepared.from_host(ram, bits[:10])
    with pytest.raises(pkg.FheRamError):
        pkg.Address.set_from_fheuint(ram, small)
