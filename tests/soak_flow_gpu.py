#!/usr/bin/env python3
"""Soak check of whole RAM operations (not collected by pytest; tests/test_gpu_soak.py runs it for a bounded time):
random RAM sizes (power-of-two and ragged row counts, 1 or 2 coordinates), word sizes, digit plans and addresses go
through read / read_prepare_write / write / read-back on the HIP path and on the oracle; outputs, rows and tree must
be bit-identical after every step.  Exercises the chain kernels, the fine split, the by-products `write` reuses, the
two streams of a write and their buffer hand-overs under changing shapes.

    python tests/soak_flow_gpu.py [seconds] [seed]        (needs an MI355X)
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

PLANS = ([3, 3, 3, 3], [4, 4, 4], [6, 6], [2, 2, 2, 2, 2, 2], [5, 4, 3], [12])


def main(seconds=120, seed=None):
    pkg = load_package()
    seed = int(time.time()) if seed is None else seed
    print("flow soak seed", seed, flush=True)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    rounds = 0
    keys_cache = {}
    while time.time() < t_end:
        plan = PLANS[int(rng.integers(0, len(PLANS)))]
        rows = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 13, 16, 32, 64, 100]))
        max_addr = rows * 4096 if rows > 1 else int(rng.choice([2, 100, 4096, 3000]))
        if rows > 1 and rng.integers(0, 3) == 0:
            max_addr -= int(rng.integers(1, 4096))            # last row partly used
        ws = int(rng.choice([1, 2, 4, 5]))
        o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws, decomp_n=plan))
        if "k" not in keys_cache:                              # keys do not depend on the RAM shape
            sk = o.secret_gen(seed & 0xFFFF)
            evk = o.evk_gen(sk, 2, 3)
            keys_cache["k"] = (sk, evk)
        sk, evk = keys_cache["k"]
        okeys = o.keys_prepare(evk)
        keys = pkg.EvaluationKeysPrepared.from_dict(evk)
        data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
        rows_ct = o.ram_encrypt(data, sk, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30)))
        ram = pkg.Ram.new_from_ram_params(ws, plan, max_addr)
        ram.load_encrypted(rows_ct)
        oram = o.ram_new()
        oram.load(rows_ct)
        for _ in range(int(rng.integers(1, 4))):
            idx = int(rng.integers(0, max_addr))
            ag = o.address_encrypt(idx, sk, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30)))
            addr, oaddr = pkg.Address(ram.params, list(ag)), o.address_new(ag)
            tag = f"round {rounds} max_addr {max_addr} ws {ws} plan {plan} idx {idx}"
            assert np.array_equal(ram.read(addr, keys), oram.read(oaddr, okeys)), "read: " + tag
            assert np.array_equal(ram.read_prepare_write(addr, keys), oram.read_prepare_write(oaddr, okeys)), "rpw: " + tag
            assert np.array_equal(ram.store_encrypted(), oram.store()), "rows after rpw: " + tag
            val = rng.integers(0, 256, size=ws, dtype=np.uint8)
            w = np.stack([o.glwe_encrypt_coeff0(int(v), sk, 20 + i, 30 + i) for i, v in enumerate(val)])
            ram.write(w, addr, keys)
            oram.write(w, oaddr, okeys)
            assert np.array_equal(ram.store_encrypted(), oram.store()), "rows after write: " + tag
            if o.p.rows > 1:
                assert np.array_equal(ram.tree(0), oram.tree(0)), "tree after write: " + tag
            back = ram.read(addr, keys)
            assert np.array_equal(back, oram.read(oaddr, okeys)), "read-back: " + tag
            for i in range(ws):
                want = o.cast_u8_to_signed(int(val[i]), 3)
                v, nz = o.glwe_decrypt(back[i], want, sk)
                assert v == want and nz < -4, (tag, v, want, nz)
        rounds += 1
        if rounds % 10 == 0:
            print(f"{rounds} RAMs", flush=True)
    print(f"flow soak ok: {rounds} RAMs in {seconds} s")
    return rounds


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else None)
