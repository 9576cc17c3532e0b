"""The parameter block of the reference's README (README.md:17-34): K_PT = 9, K_EVK = 5 * BASEK for every evaluation key
— 5-limb trace / packing keys instead of the source constants' 4 (parameters.rs:17).  The published 450 ms / 1200 ms were
taken with this block (README.md:36), so the bench can be run on it (`bench.py --params readme`); here every op the
5-limb keys reach is compared with the oracle bit for bit, and the example flow runs at MAX_ADDR = 2^14 (the committed
digests at 2^14 and 2^18 are reproduced by tests/test_gpu_golden.py)."""
import numpy as np
import pytest

from test_gpu_parity import World, full_flow, rand_glwe

pytestmark = pytest.mark.gpu
README = {"k_glwe_pt": 9, "k_evk_trace": 85}


@pytest.fixture(scope="module")
def wr(po):
    return World(po, 1 << 14, word_size=4, seed=51, **README)


def test_layouts_follow_the_readme_block(wr):
    p = wr.ram.params
    assert (p.k_glwe_pt(), p.k_evk_trace(), p.k_evk_ggsw_inv()) == (9, 85, 85)
    assert wr.o.p.atk_trace_len == 3 * 5 * 2 * 4096          # evk_glwe_infos (parameters.rs:71-81): dnum_ct rows of 5 limbs
    assert all(k.size == wr.o.p.atk_trace_len for k in wr.keys.atk_glwe)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_automorphism_family_with_5_limb_keys(wr, mode):
    a = rand_glwe(wr.rng, 3, edge=True)
    for gi in (0, 1, 7, 11):
        gal = int(wr.evk["gal_els"][gi])
        got = wr.ram.glwe_automorphism(wr.keys, gal, mode, a)
        for i in range(a.shape[0]):
            assert np.array_equal(got[i], wr.o.glwe_automorphism(wr.okeys, gal, mode, a[i])), (gal, mode, i)


@pytest.mark.parametrize("start,end,batch", [(0, 12, 3), (0, 12, 40), (0, 12, 300), (2, 9, 300), (11, 12, 1)])
def test_trace_with_5_limb_keys_every_launch_shape(wr, start, end, batch):
    """batch 3: single-launch chain with in-kernel hand-offs (30 workgroups per ciphertext) / fine split; 40: limb
    parallel; 300: fused chain kernel"""
    a = rand_glwe(wr.rng, batch, edge=batch > 2)
    got = wr.ram.glwe_trace(wr.keys, start, end, a)
    for i in sorted(set([0, 1 % batch, 2 % batch, batch // 2, batch - 1])):
        assert np.array_equal(got[i], wr.o.glwe_trace(wr.okeys, start, end, a[i])), (start, end, batch, i)


@pytest.mark.parametrize("count", [2, 5, 13, 70])
def test_pack_with_5_limb_keys(wr, po, count):
    cts = rand_glwe(wr.rng, count, edge=count > 3)
    present = np.zeros(4096, dtype=np.uint8)
    order = []
    for j in range(4096):
        jr = int(po.lib().fo_reverse_bits_msb(j, 12))
        if jr < count:
            present[j] = 1
            order.append(jr)
    assert np.array_equal(wr.ram.glwe_pack(wr.keys, cts), wr.o.glwe_pack(wr.okeys, cts[order], present))


def test_ram_flow_2_14_readme_block(wr):
    full_flow(wr)
    st = wr.ram.tail_stats()
    assert st["launches"] > 0 and st["fallbacks"] == 0, st     # 30 workgroups per ciphertext still sit side by side on one XCD


def test_key_layout_mismatch_is_refused(wr, po):
    """4-limb trace keys handed to a context built for 5-limb ones (and the reverse) must not be read past their end"""
    pkg = wr.pkg
    other = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)           # source constants
    with pytest.raises(pkg.FheRamError) as e:
        other._use_keys(wr.keys)
    assert e.value.code == 1
    o4 = po.Oracle(po.OParams(max_addr=1 << 12))
    k4 = pkg.EvaluationKeysPrepared.from_dict(o4.evk_gen(o4.secret_gen(1), 2, 3))
    with pytest.raises(pkg.FheRamError):
        wr.ram._use_keys(k4)
