"""Host-side pieces of the setup mirror (fhe-ram_amd/api.py) against the oracle, no GPU needed:
cast_u8_to_signed (examples/fhe-ram.rs:25-32), the plaintext encoding and the noise scale the host
sampler is asked for, and the draw counts of each encrypt_sk call."""
import numpy as np
import pytest

from _pkg import load_package


def test_cast_u8_to_signed_matches_reference_semantics(po):
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=1 << 12))
    for bits in (1, 3, 5, 8):
        for v in range(256):
            assert pkg.cast_u8_to_signed(v, bits) == o.cast_u8_to_signed(v, bits), (v, bits)
    assert pkg.cast_u8_to_signed(0b101, 3) == -3 and pkg.cast_u8_to_signed(0b011, 3) == 3   # :25-32


@pytest.mark.parametrize("k", [3, 8, 17, 20])
def test_encode_coeff_matches_oracle_word_encryption(po, k):
    """encode_coeff(v, k) is what encrypt_glwe puts on coefficient 0: the oracle's phase of a noiseless,
    maskless encryption shows the same limbs"""
    pkg = load_package()
    size = -(-k // 17)
    for v in (0, 1, 5, 127, 128, 200, 255, -1, -128):
        limbs = pkg.encode_coeff(v, k)
        assert len(limbs) == size and all(-(1 << 16) <= d < (1 << 16) for d in limbs)
        val = sum(d << (17 * (size - 1 - j)) for j, d in enumerate(limbs))
        want = (v << (size * 17 - k))
        assert (val - want) % (1 << (17 * size)) == 0        # same torus element


def test_noise_scale_and_draw_counts(po):
    pkg = load_package()
    assert pkg.noise_scale(51) == 1.0 and pkg.noise_scale(68) == 1.0 and pkg.noise_scale(85) == 1.0
    assert pkg.noise_scale(40) == float(1 << 11) and pkg.noise_scale(60) == float(1 << 8)

    class Counting:
        def __init__(self):
            self.u, self.g = [], []

        def uniform_limbs(self, count):
            self.u.append(count)
            return np.zeros(count, dtype=np.int64)

        def gaussian(self, count, scale=1.0):
            self.g.append((count, scale))
            return np.zeros(count, dtype=np.int64)

    class FakeRam:   # only what the mirror touches before it reaches the C ABI
        def __init__(self, params):
            self.params, self.n_shards, self.shard, self._h = params, 1, 0, None

        def _chk(self, rc):
            raise RuntimeError("stop before the C ABI")

    p = pkg.Parameters(max_addr=1 << 18)
    xa, xe = Counting(), Counting()

    class Sk:
        _h = None
    with pytest.raises(Exception):
        pkg.Ram.encrypt_sk(FakeRam(p), np.zeros((1 << 18) * 4, dtype=np.uint8), Sk(), xa, xe)
    # ram.rs:358-379: word_size x rows ciphertexts of 3 limbs, one noise polynomial each
    assert xa.u == [4 * 64 * 3 * 4096] and xe.g == [(4 * 64 * 4096, 1.0)]
