"""Setup side on the device against the oracle's setup side: with the same draws
(the oracle's seeded `Source` replayed on the host and handed over), device encryption of GLWEs, of
the RAM, of an address and of the evaluation keys is bit-exact, decryption matches, and a RAM that was
set up entirely on the device runs the reference's example flow (examples/fhe-ram.rs:34-177)."""
import numpy as np
import pytest

from _pkg import load_package

pytestmark = pytest.mark.gpu
N = 4096


@pytest.fixture(scope="module")
def env(po):
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=1 << 14))
    sk = o.secret_gen(7)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
    return pkg, o, sk, ram, pkg.GLWESecret(ram, sk)


@pytest.mark.parametrize("size,k,pt_col", [(3, 51, 0), (4, 68, 0), (4, 68, 1), (5, 85, 0), (3, 40, 0), (4, 60, 1)])
def test_glwe_encrypt_bit_exact(env, size, k, pt_col):
    """every ciphertext shape of the path (RAM rows / GGSW rows / key rows), plus precisions that are not
    a multiple of base2k (noise scaled by 2^((limb+1)*base2k-k))"""
    pkg, o, sk, ram, dsk = env
    rng = np.random.default_rng(size * 100 + k)
    pt = rng.integers(-(1 << 16), 1 << 16, size=(2, N), dtype=np.int64)
    pt[0, :4] = [-(1 << 16), (1 << 16) - 1, 0, 1]
    want = o.glwe_encrypt_sk(size, k, pt, pt_col, sk, 11, 12)
    got = ram.glwe_encrypt_sk(dsk, 1, size, k, pt[None], pt_col, o.source(11), o.source(12))
    assert np.array_equal(got[0], want)
    want0 = o.glwe_encrypt_sk(size, k, None, 0, sk, 13, 14)          # encryption of zero
    got0 = ram.glwe_encrypt_sk(dsk, 1, size, k, None, 0, o.source(13), o.source(14))
    assert np.array_equal(got0[0], want0)


def test_glwe_decrypt_bit_exact_and_batch(env):
    pkg, o, sk, ram, dsk = env
    rng = np.random.default_rng(3)
    for size in (3, 4, 5):
        cts = rng.integers(-(1 << 16), 1 << 16, size=(5, size * 2 * N), dtype=np.int64)
        got = ram.glwe_decrypt(dsk, cts, size)
        for i in range(5):
            assert np.array_equal(got[i], o.glwe_phase(cts[i], sk)), (size, i)


def test_word_encrypt_decrypt_roundtrip(env):
    """encrypt_glwe / decrypt_glwe of the example (examples/fhe-ram.rs:179-237) on the device"""
    pkg, o, sk, ram, dsk = env
    vals = [0, 1, 5, 7, 200, 255]
    cts = ram.encrypt_word(dsk, vals, o.source(21), o.source(22))
    xa, xe = 21, 22   # the oracle encrypts one word per call from fresh sources: compare the first
    assert np.array_equal(cts[0], o.glwe_encrypt_coeff0(vals[0], sk, xa, xe))
    wants = [pkg.cast_u8_to_signed(v, 3) for v in vals]
    assert wants == [o.cast_u8_to_signed(v, 3) for v in vals]
    for (v, noise), want, ct in zip(ram.decrypt_coeff(dsk, cts, wants), wants, cts):
        ov, onoise = o.glwe_decrypt(ct, want, sk)
        assert v == want == ov and noise == pytest.approx(onoise, abs=1e-9) and noise < -4.0


@pytest.mark.parametrize("max_addr", [1 << 12, 1 << 14, 3 << 12, (1 << 14) - 5])
def test_ram_encrypt_bit_exact(po, max_addr):
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=max_addr))
    sk = o.secret_gen(8)
    data = np.random.default_rng(max_addr).integers(0, 256, size=max_addr * 4, dtype=np.uint8)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], max_addr)
    ram.encrypt_sk(data, pkg.GLWESecret(ram, sk), o.source(31), o.source(32))
    assert np.array_equal(ram.store_encrypted(), o.ram_encrypt(data, sk, 31, 32))
    with pytest.raises(pkg.FheRamError, match="invalid data"):
        ram.encrypt_sk(data[:-4], pkg.GLWESecret(ram, sk), o.source(31), o.source(32))
    with pytest.raises(pkg.FheRamError, match="invalid data"):
        ram.encrypt_sk(data[:-1], pkg.GLWESecret(ram, sk), o.source(31), o.source(32))


def test_ram_encrypt_sharded_rows(po):
    pkg = load_package()
    max_addr = 1 << 14
    o = po.Oracle(po.OParams(max_addr=max_addr))
    sk = o.secret_gen(8)
    data = np.random.default_rng(1).integers(0, 256, size=max_addr * 4, dtype=np.uint8)
    want = o.ram_encrypt(data, sk, 31, 32)
    for shard in range(2):
        ram = pkg.Ram(pkg.Parameters(max_addr=max_addr), shard=shard, n_shards=2)
        ram.encrypt_sk(data, pkg.GLWESecret(ram, sk), o.source(31), o.source(32))
        assert np.array_equal(ram.store_encrypted(), want[:, shard::2])


@pytest.mark.parametrize("value", [0, 1, 4095, 4096, 12345, (1 << 14) - 1])
def test_address_encrypt_bit_exact(env, value):
    pkg, o, sk, ram, dsk = env
    addr = pkg.Address.encrypt_sk(ram, value, dsk, o.source(41), o.source(42))
    want = o.address_encrypt(value, sk, 41, 42)
    assert np.array_equal(np.stack(addr.digits), want)


def test_keys_encrypt_bit_exact(env):
    pkg, o, sk, ram, dsk = env
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, dsk, o.source(51), o.source(52), keep_std=True)
    want = o.evk_gen(sk, 51, 52)
    assert list(keys.gal_els) == list(want["gal_els"])
    for i in range(12):
        assert np.array_equal(keys.atk_glwe[i], want["atk_glwe"][i].ravel()), i
    assert np.array_equal(keys.tsk_ggsw_inv, want["tsk"].ravel())
    assert np.array_equal(keys.atk_ggsw_inv, want["atk_ggsw_inv"].ravel())


def test_setup_argument_errors(env, po):
    pkg, o, sk, ram, dsk = env
    with pytest.raises(pkg.FheRamError, match=r"\{-1, 0, 1\}"):
        pkg.GLWESecret(ram, np.full(N, 2, dtype=np.int64))
    other = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
    with pytest.raises(pkg.FheRamError, match="another context"):
        other.glwe_encrypt_sk(dsk, 1, 3, 51, None, 0, o.source(1), o.source(2))

    class Bad:
        def uniform_limbs(self, count):
            return np.full(count, 1 << 16, dtype=np.int64)

        def gaussian(self, count, scale):
            return np.zeros(count, dtype=np.int64)
    with pytest.raises(pkg.FheRamError, match="mask limb"):
        ram.glwe_encrypt_sk(dsk, 1, 3, 51, None, 0, Bad(), Bad())
    with pytest.raises(pkg.FheRamError, match="does not fit"):
        ram.glwe_encrypt_sk(dsk, 1, 3, 60, None, 0, o.source(1), o.source(2))
    with pytest.raises(pkg.FheRamError, match=r"base2d.max\(\) > value"):      # address.rs:98
        pkg.Address.encrypt_sk(ram, 1 << 14, dsk, o.source(1), o.source(2))
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, dsk, o.source(1), o.source(2))
    other.load_encrypted(np.zeros((4, 1, other.params.glwe_len()), dtype=np.int64))
    with pytest.raises(pkg.FheRamError, match="another context's device"):         # device-only keys cannot move
        other.read(pkg.Address.alloc_from_params(other.params), keys)


@pytest.mark.parametrize("max_addr", [1 << 14, 1 << 16, 1 << 21, 1 << 24])
def test_example_flow_with_device_setup(po, max_addr):
    """examples/fhe-ram.rs:34-177 with every setup step on the device (host-side sources): keys, RAM and
    address are never seen by the host; only the words read back are decrypted (on the device too)."""
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=max_addr))
    p = o.p
    sk = o.secret_gen(0)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], max_addr)
    dsk = pkg.GLWESecret(ram, sk)
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, dsk, o.source(1), o.source(2))
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=max_addr * 4, dtype=np.uint8)
    ram.encrypt_sk(data, dsk, o.source(3), o.source(4))
    idx = int(rng.integers(0, max_addr))
    addr = pkg.Address.encrypt_sk(ram, idx, dsk, o.source(5), o.source(6))

    def check(cts, data):
        wants = [pkg.cast_u8_to_signed(int(data[i + 4 * idx]), p.k_glwe_pt) for i in range(4)]
        for (v, noise), want in zip(ram.decrypt_coeff(dsk, cts, wants), wants):
            assert v == want and noise < -(p.k_glwe_pt + 1.0), (v, want, noise)

    check(ram.read(addr, keys), data)
    check(ram.read_prepare_write(addr, keys), data)
    value = rng.integers(0, 256, size=4, dtype=np.uint8)
    ram.write(ram.encrypt_word(dsk, value, o.source(7), o.source(8)), addr, keys)
    data[4 * idx:4 * idx + 4] = value
    check(ram.read(addr, keys), data)
    if max_addr > (1 << 16):      # BASELINE.json configs[4] (2^21) and the largest RAM the digit plan allows (N^2 = 2^24):
        return                    # the flow's own assertions are the check; the oracle would take minutes
    # and the device-made setup is the oracle's setup: the oracle reads the same word from the same state
    okeys = o.keys_prepare(o.evk_gen(sk, 1, 2))
    oram = o.ram_new()
    oram.load(ram.store_encrypted())
    got = oram.read(o.address_new(np.stack(addr.digits)), okeys)
    assert np.array_equal(got, ram.read(addr, keys))


def test_cpp_host_mirror_runs_device_setup_flow(tmp_path):
    """fhe-ram_amd/host/host_check.cpp: the C++ mirror sets up secret, keys, RAM and address on the device
    (noiseless sampler), reads a word and decrypts it."""
    import os
    import shutil
    import subprocess
    pkg = load_package()
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_check")
    libdir = os.path.dirname(pkg.library_path())
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-o", exe, os.path.join(root, "fhe-ram_amd", "host", "host_check.cpp"),
                           "-L" + libdir, "-lfheram", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "device-side setup + read + decrypt: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("max_addr,ws,decomp", [(2, 1, (3, 3, 3, 3)), (1000, 1, (3, 3, 3, 3)), (5000, 3, (3, 3, 3, 3)), (1 << 13, 8, (3, 3, 3, 3)),
                                                (4097, 2, (3, 3, 3, 3)), (1 << 14, 2, (4, 4, 4)), (3 << 12, 1, (2, 2, 2, 2, 2, 2)), (1 << 13, 1, (6, 6)),
                                                (1 << 13, 1, (12,)), (5000, 2, (5, 4, 3))])
def test_odd_shapes_flow_and_oracle_agreement(po, max_addr, ws, decomp):
    """RAM sizes that are not powers of two (ragged last row, digit plans with a short last digit), below N, and
    word sizes other than 4 (ram.rs:72-87 takes any word_size / max_addr): device setup, the example's assertions,
    other digit plans (DECOMP_N, parameters.rs:18,168), and bit-exact agreement of read / read_prepare_write /
    write with the oracle run on the same state."""
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws, decomp_n=decomp))
    p = o.p
    sk = o.secret_gen(3)
    ram = pkg.Ram.new_from_ram_params(ws, list(decomp), max_addr)
    dsk = pkg.GLWESecret(ram, sk)
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, dsk, o.source(1), o.source(2), keep_std=True)
    rng = np.random.default_rng(max_addr + ws)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    ram.encrypt_sk(data, dsk, o.source(3), o.source(4))
    okeys = o.keys_prepare({"gal_els": keys.gal_els, "atk_glwe": np.stack(keys.atk_glwe), "atk_ggsw_inv": keys.atk_ggsw_inv, "tsk": keys.tsk_ggsw_inv})
    oram = o.ram_new()
    oram.load(ram.store_encrypted())
    for idx in sorted({0, max_addr - 1, int(rng.integers(0, max_addr))}):
        addr = pkg.Address.encrypt_sk(ram, idx, dsk, o.source(5 + idx), o.source(6 + idx))
        oaddr = o.address_new(np.stack(addr.digits))
        wants = [pkg.cast_u8_to_signed(int(data[i + ws * idx]), p.k_glwe_pt) for i in range(ws)]
        got = ram.read(addr, keys)
        assert np.array_equal(got, oram.read(oaddr, okeys))
        for (v, noise), want in zip(ram.decrypt_coeff(dsk, got, wants), wants):
            assert v == want and noise < -(p.k_glwe_pt + 1.0), (idx, v, want, noise)
        assert np.array_equal(ram.read_prepare_write(addr, keys), oram.read_prepare_write(oaddr, okeys))
        value = rng.integers(0, 256, size=ws, dtype=np.uint8)
        w = ram.encrypt_word(dsk, value, o.source(7), o.source(8))
        ram.write(w, addr, keys)
        oram.write(w, oaddr, okeys)
        data[ws * idx:ws * idx + ws] = value
        assert np.array_equal(ram.store_encrypted(), oram.store())
