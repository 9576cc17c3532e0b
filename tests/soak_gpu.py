#!/usr/bin/env python3
"""Soak check of the HIP path (tests/test_gpu_soak.py runs it for a bounded time in the driver's suite;
longer runs by hand): for a given number of seconds, random
ciphertext batches of the sizes that select each launch decomposition (fused, column split, limb
parallel) go through trace steps, automorphisms, packing and external products, and a random sample of
every result is compared bit for bit with the oracle; the same GPU call is also repeated and must
reproduce itself exactly (the kernels are deterministic: any difference is a synchronisation bug).

    python tests/soak_gpu.py [seconds]        (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from _pkg import load_package  # noqa: E402


def main(seconds=120, seed=None):
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=1 << 14))
    sk = o.secret_gen(1)
    evk = o.evk_gen(sk, 2, 3)
    okeys = o.keys_prepare(evk)
    keys = pkg.EvaluationKeysPrepared.from_dict(evk)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
    seed = int(time.time()) if seed is None else seed
    print("soak seed", seed, flush=True)
    rng = np.random.default_rng(seed)
    glen = o.p.glwe_len
    t_end = time.time() + seconds
    rounds = checks = 0
    while time.time() < t_end:
        # batch sizes that select every launch decomposition: fine limb split (<= 10 ciphertexts), limb
        # parallel (<= 32), split by column (<= 128), fused
        batch = int(rng.choice([1, 3, 4, 8, 10, 11, 16, 32, 33, 48, 64, 65, 100, 128, 256, 300, 512]))
        a = rng.integers(-(1 << 16), 1 << 16, size=(batch, glen), dtype=np.int64)
        sample = rng.choice(batch, size=min(batch, 3), replace=False)
        kind = int(rng.integers(0, 4))
        if kind == 0:      # trace steps (KS_TRACE)
            s = int(rng.integers(0, 12)); e = int(rng.integers(s + 1, 13))
            e = min(e, s + 2)
            got = ram.glwe_trace(keys, s, e, a)
            again = ram.glwe_trace(keys, s, e, a)
            ref = lambda i: o.glwe_trace(okeys, s, e, a[i])
        elif kind == 1:    # automorphism family (KS_AUTO / ADD / SUBNEG)
            mode = int(rng.integers(0, 3)); g = int(evk["gal_els"][int(rng.integers(0, 12))])
            got = ram.glwe_automorphism(keys, g, mode, a)
            again = ram.glwe_automorphism(keys, g, mode, a)
            ref = lambda i: o.glwe_automorphism(okeys, g, mode, a[i])
        elif kind == 2:    # external product
            ggsw = rng.integers(-(1 << 16), 1 << 16, size=o.p.ggsw_len, dtype=np.int64)
            got = ram.glwe_external_product(a, ggsw)
            again = ram.glwe_external_product(a, ggsw)
            ref = lambda i: o.glwe_external_product(a[i], ggsw)
        else:              # packing tree (KS_PAIR + KS_TRACE), one output
            count = int(rng.choice([2, 3, 5, 8, 16, 31, 64]))
            a = a[:1].repeat(count, axis=0) + rng.integers(-3, 4, size=(count, glen), dtype=np.int64)
            a = np.clip(a, -(1 << 16), (1 << 16) - 1)
            got = ram.glwe_pack(keys, a)[None]
            again = ram.glwe_pack(keys, a)[None]
            present = np.zeros(o.p.n, dtype=np.uint8)      # the RAM's bit-reversed feed order (ram.rs:425-444)
            order = []
            for j in range(o.p.n):
                jr = int(po.lib().fo_reverse_bits_msb(j, o.p.log_n))
                if jr < count:
                    present[j] = 1
                    order.append(jr)
            ref = lambda i: o.glwe_pack(okeys, a[order], present)
            sample = [0]
        assert np.array_equal(got, again), f"round {rounds}: kind {kind} batch {batch} is not reproducible"
        for i in sample:
            assert np.array_equal(got[i], ref(i)), f"round {rounds}: kind {kind} batch {batch} ct {i} differs from the oracle"
            checks += 1
        rounds += 1
        if rounds % 50 == 0:
            print(f"{rounds} rounds, {checks} oracle checks", flush=True)
    print(f"soak ok: {rounds} rounds, {checks} oracle checks in {seconds} s")
    return rounds, checks


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 120)
