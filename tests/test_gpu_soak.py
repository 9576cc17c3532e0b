"""Bounded soak of the barrier-trimmed transforms inside the driver-run GPU suite (tests/soak_gpu.py):
random batches of every size class (each selects another launch decomposition) through trace steps,
the automorphism family, external products and the packing tree; every call is repeated and must
reproduce itself bit for bit (a difference is a synchronisation bug, the class fixed in 6ea3fe6), and a
sample of every result is compared with the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_soak_every_launch_decomposition_reproducible_and_exact(po):
    import soak_gpu
    rounds, checks = soak_gpu.main(seconds=25, seed=20261003)
    assert rounds >= 20 and checks >= 40
