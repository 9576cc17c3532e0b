"""Row-sharded HIP path on ONE GPU: G shard contexts in one process, exchange through host buffers.
Must be bit-identical to the unsharded oracle (and therefore to the unsharded HIP path)."""
import numpy as np
import pytest

from _pkg import load_package

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_shards,max_addr", [(2, 4 * 4096), (4, 4 * 4096), (4, 16 * 4096)])
def test_sharded_flow_bit_exact(po, n_shards, max_addr):
    pkg = load_package()
    ws = 2
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
    sk = o.secret_gen(5)
    evk = o.evk_gen(sk, 6, 7)
    okeys = o.keys_prepare(evk)
    keys = pkg.EvaluationKeysPrepared.from_dict(evk)
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    rows = o.ram_encrypt(data, sk, 9, 10)
    idx = int(rng.integers(0, max_addr))
    addr_g = o.address_encrypt(idx, sk, 11, 12)
    oaddr = o.address_new(addr_g)
    oram = o.ram_new()
    oram.load(rows)

    params = pkg.Parameters(max_addr=max_addr, word_size=ws)
    shards = [pkg.Ram(params, 0, shard=g, n_shards=n_shards) for g in range(n_shards)]
    addrs = [pkg.Address(params, list(addr_g)) for _ in shards]
    for g, r in enumerate(shards):
        r.load_encrypted(rows[:, g::n_shards])

    def read(prepare_write):
        partials = np.stack([r.read_partial(a, keys, prepare_write) for r, a in zip(shards, addrs)])   # "all-gather"
        return shards[0].read_finish(addrs[0], keys, partials, prepare_write)

    assert np.array_equal(read(False), oram.read(oaddr, okeys))
    assert np.array_equal(read(True), oram.read_prepare_write(oaddr, okeys))
    full = oram.store()
    for g, r in enumerate(shards):
        assert np.array_equal(r.store_encrypted(), full[:, g::n_shards])
    assert np.array_equal(shards[0].tree(0), oram.tree(0))
    val = rng.integers(0, 256, size=ws, dtype=np.uint8)
    w = np.stack([o.glwe_encrypt_coeff0(int(v), sk, 20 + i, 30 + i) for i, v in enumerate(val)])
    ct_lo = shards[0].write_root(w, addrs[0], keys)                                                   # "broadcast"
    for r, a in zip(shards, addrs):
        r.write_shard(a, keys, ct_lo)
    oram.write(w, oaddr, okeys)
    full = oram.store()
    for g, r in enumerate(shards):
        assert np.array_equal(r.store_encrypted(), full[:, g::n_shards])
    assert np.array_equal(shards[0].tree(0), oram.tree(0))
    back = read(False)
    assert np.array_equal(back, oram.read(oaddr, okeys))
    for i in range(ws):
        want = o.cast_u8_to_signed(int(val[i]), 3)
        v, nz = o.glwe_decrypt(back[i], want, sk)
        assert v == want and nz < -4
    # misuse: the unsharded entry points refuse a sharded context
    with pytest.raises(pkg.FheRamError):
        shards[0].read(addrs[0], keys)
