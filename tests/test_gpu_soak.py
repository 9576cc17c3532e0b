"""Bounded soak of the barrier-trimmed transforms inside the driver-run GPU suite (tests/soak_gpu.py):
random batches of every size class (each selects another launch decomposition) through trace steps,
the automorphism family, external products and the packing tree; every call is repeated and must
reproduce itself bit for bit (a difference is a synchronisation bug, the class fixed in 6ea3fe6), and a
sample of every result is compared with the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_soak_every_launch_decomposition_reproducible_and_exact(po):
    import soak_gpu
    rounds, checks = soak_gpu.main(seconds=12, seed=20261003)
    assert rounds >= 20 and checks >= 40


def test_soak_whole_ram_flows_over_random_shapes(po):
    """tests/soak_flow_gpu.py for a bounded time: random RAM sizes (ragged rows, 1 and 2 coordinates), word sizes, digit
    plans and addresses through read / read_prepare_write / write / read-back, every output and the whole state
    bit-identical to the oracle's."""
    import soak_flow_gpu
    assert soak_flow_gpu.main(seconds=20, seed=20261004) >= 2


def test_soak_single_launch_trace_chain(po):
    """tests/soak_tail_gpu.py for a bounded time: random batches of 1..8 ciphertexts through random trace ranges in ONE
    launch each (in-kernel hand-offs), three times each (must reproduce itself) and against the oracle, while another
    host thread fills the chip from a second context."""
    import soak_tail_gpu
    rounds, st = soak_tail_gpu.main(seconds=8, seed=20261005)
    assert rounds >= 5 and st["launches"] >= 3 * rounds and st["fallbacks"] <= st["launches"]


def test_soak_mid_batch_chains(po):
    """The same for k_chain_mid: random batches of 9..64 ciphertexts (12 / 8 / 4 workgroups per ciphertext on one XCD,
    in-kernel hand-offs, giving up per ciphertext) while a second context competes for the CUs: reproducible, exact, and
    whatever gave up was redone."""
    import soak_tail_gpu
    rounds, st = soak_tail_gpu.main(seconds=8, seed=20261006, batches=(9, 64))
    assert rounds >= 5 and st["launches"] >= 3 * rounds


@pytest.mark.parametrize("params", ["source", "readme"])
def test_noise_growth_over_consecutive_write_cycles(po, params):
    """BASELINE.json configs[3] 'noise growth checked': a bounded run of tests/noise_growth_gpu.py (the long one is
    profiles/r03_noise_growth*.json): 400 read_prepare_write + write cycles at 2^14; every sampled read decrypts to the
    plaintext model with noise below the reference's bound, and the fitted growth leaves room for the README's 40
    million cycles."""
    import noise_growth_gpu
    out = noise_growth_gpu.run(cycles=400, sample_every=50, log_max_addr=14, params=params, pool=16)
    assert out["worst_noise_bits"] < out["noise_bound_bits"]
    assert out["last"]["cycle"] == 400 and len(out["trajectory"]) == 9
    assert out["last"]["never_written_mean_bits"] > out["first"]["never_written_mean_bits"]      # this is the noise that accumulates
    print({k: out[k] for k in ("worst_noise_bits", "noise_bound_bits", "cycles_per_s", "fit_variance")})
