#!/usr/bin/env python3
"""Soak check of the single-launch trace chain (k_trace_tail: workgroups of a ciphertext hand over inside the kernel,
through the L2 of the XCD they share): for a given number of seconds, random batches of 1..8 ciphertexts go through
random trace ranges of 2..12 steps; every call is made three times and must reproduce itself bit for bit (a stale read
after a hand-off would show here), one ciphertext per round is compared with the oracle, and a second host thread keeps
another context busy with chip-filling launches so that the groups assemble under changing CU availability.  At the
end the launch / fallback counters are printed (fallbacks are legal: their results are checked like any other).

    python tests/soak_tail_gpu.py [seconds] [seed] [min_batch max_batch]       (needs an MI355X; 9 64: the same for k_chain_mid)
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from _pkg import load_package  # noqa: E402


def main(seconds=120, seed=None, disturb=True, batches=(1, 8)):
    """batches: inclusive range of the batch sizes drawn: (1, 8) = k_trace_tail's territory, (9, 64) = k_chain_mid's"""
    pkg = load_package()
    o = po.Oracle(po.OParams(max_addr=1 << 14))
    sk = o.secret_gen(1)
    evk = o.evk_gen(sk, 2, 3)
    okeys = o.keys_prepare(evk)
    keys = pkg.EvaluationKeysPrepared.from_dict(evk)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
    seed = int(time.time()) if seed is None else seed
    print("tail soak seed", seed, flush=True)
    rng = np.random.default_rng(seed)
    glen = o.p.glwe_len
    stop = threading.Event()
    errs = []

    def noise():
        try:
            other = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
            okeys2 = pkg.EvaluationKeysPrepared.from_dict(evk)
            r2 = np.random.default_rng(seed + 1)
            g = int(evk["gal_els"][3])
            while not stop.is_set():
                b = int(r2.choice([64, 256, 300]))
                other.glwe_automorphism(okeys2, g, 0, r2.integers(-(1 << 16), 1 << 16, size=(b, glen), dtype=np.int64))
                time.sleep(float(r2.uniform(0.0, 0.002)))
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=noise) if disturb else None
    if th:
        th.start()
    t_end = time.time() + seconds
    rounds = checks = 0
    try:
        while time.time() < t_end and not errs:
            batch = int(rng.integers(batches[0], batches[1] + 1))
            s = int(rng.integers(0, 11))
            e = int(rng.integers(s + 2, 13))
            a = rng.integers(-(1 << 16), 1 << 16, size=(batch, glen), dtype=np.int64)
            if rounds % 7 == 0:      # digit extremes
                a[0, :] = -(1 << 16)
                a[-1, ::2] = (1 << 16) - 1
            got = ram.glwe_trace(keys, s, e, a)
            for rep in range(2):
                assert np.array_equal(got, ram.glwe_trace(keys, s, e, a)), f"round {rounds}: batch {batch} steps {s}..{e} is not reproducible"
            i = int(rng.integers(0, batch))
            assert np.array_equal(got[i], o.glwe_trace(okeys, s, e, a[i])), f"round {rounds}: batch {batch} steps {s}..{e} ct {i} differs from the oracle"
            checks += 1
            rounds += 1
            if rounds % 50 == 0:
                print(rounds, "rounds,", checks, "oracle checks", flush=True)
    finally:
        stop.set()
        if th:
            th.join()
    assert not errs, errs
    st = ram.tail_stats() if batches[1] <= 8 else ram.mid_stats()
    print(f"tail soak ok: {rounds} rounds ({3 * rounds} launches), {checks} oracle checks in {seconds} s; {st}", flush=True)
    return rounds, st


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else None,
         batches=(int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 8))
