#!/usr/bin/env python3
"""Counterpart of the reference's only driver, /root/reference/examples/fhe-ram.rs:34-177 (SURVEY.md
§8(f) row N1), on the MI355X evaluator: key generation, RAM and address encryption (setup side: the
oracle, standing in for Poulpy on the host), then timed `read`, `read_prepare_write`, `write`, and
the reference's own assertions after each: decrypted coefficient 0 == cast_u8_to_signed(data) and
noise < -(k_pt+1).  Lives under tests/ because the setup side is the test-only oracle.

    python tests/example_flow.py [log2(max_addr)]        (needs an MI355X)
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


def main(log_max_addr=14):
    pkg = load_package()
    print("Starting!")                                                     # examples/fhe-ram.rs:35
    max_addr = 1 << log_max_addr
    o = po.Oracle(po.OParams(max_addr=max_addr))                           # Parameters::new()            :46
    p = o.p
    sk = o.secret_gen(0)                                                   # fill_ternary_prob(0.5)       :49-50
    evk = o.evk_gen(sk, 0, 0)                                              # EvaluationKeys::encrypt_sk   :52-53
    keys = pkg.EvaluationKeysPrepared.from_dict(evk)                       # EvaluationKeysPrepared       :61-63
    rng = np.random.default_rng(5)                                         # Source::new([5u8; 32])       :66
    ws = p.word_size
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)        # :72-73
    ram = pkg.Ram.new_from_ram_params(ws, p.decomp_n, max_addr)            # Ram::new()                   :76
    ram.load_encrypted(o.ram_encrypt(data, sk, 1, 2))                      # ram.encrypt_sk               :79
    idx = int(rng.integers(0, max_addr))                                   # :85
    addr = pkg.Address(ram.params, list(o.address_encrypt(idx, sk, 3, 4)))  # addr.encrypt_sk             :88-95

    def check(ct, data):                                                   # :104-115
        for i in range(ws):
            want = o.cast_u8_to_signed(int(data[i + ws * idx]), p.k_glwe_pt)
            value, noise = o.glwe_decrypt(ct[i], want, sk)
            assert value == want, (value, want)
            print(f"noise: {noise}")
            assert noise < -(p.k_glwe_pt + 1.0), f"{noise} >= {p.k_glwe_pt + 1.0}"

    for _ in range(3):                                                     # first calls upload keys / address, wake the clocks
        ram.read(addr, keys)
    t = time.perf_counter()
    ct = ram.read(addr, keys)                                              # :98-101
    print(f"READ Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    check(ct, data)
    t = time.perf_counter()
    ct = ram.read_prepare_write(addr, keys)                                # :118-124
    print(f"READ_PREPARE_WRITE Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    check(ct, data)
    value = rng.integers(0, 256, size=ws, dtype=np.uint8)                  # :141-142
    ct_w = np.stack([o.glwe_encrypt_coeff0(int(v), sk, 10 + i, 20 + i) for i, v in enumerate(value)])   # :145-148
    t = time.perf_counter()
    ram.write(ct_w, addr, keys)                                            # :151-154
    ram.sync()
    print(f"WRITE Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    for i in range(ws):                                                    # :157-159
        data[i + ws * idx] = value[i]
    check(ram.read(addr, keys), data)                                      # :162-176
    print("ok:", ram.device_info())


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 14)
